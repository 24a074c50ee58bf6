"""Minimal stand-in for the parts of ``lmfit`` that ExTrack's fitting path uses.

The reference builds ``lmfit.Parameters`` (extrack/tracking.py:1210,1287) and calls
``lmfit.minimize(cum_Proba_Cs, params, args=..., method=..., nan_policy='propagate')``
(extrack/tracking.py:1371).  lmfit is a third-party dependency (requirements.txt: ``lmfit>=0.9.7``)
that is neither vendored in the reference nor installed in this image, so:

  * if ``import lmfit`` works, it is used unchanged (``HAVE_LMFIT = True``);
  * otherwise the classes below provide the same surface: ``Parameters.add(name, value, vary, min, max,
    expr, brute_step)``, ``params[name].value/.vary/.min/.max/.expr``, ``minimize(...)`` returning an object
    with ``.params``, ``.residual``, ``.nfev``, ``.success``, ``.message`` (consumers:
    ExTrack_GUI.py:304,321,341-343).

Parity note: the optimiser TRAJECTORY of real lmfit is unpinned (no fixture exists in the reference).
What is reproduced from lmfit's published behaviour: scalar methods minimise the objective's scalar
return directly; bounded parameters are mapped to unbounded internal variables with the MINUIT
transforms (sin for two-sided, sqrt for one-sided bounds); ``expr`` parameters are re-evaluated after
every update.
"""
import ast
import copy
import math
from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - lmfit is absent in the build image
    import lmfit as _lmfit
    HAVE_LMFIT = True
except ImportError:
    _lmfit = None
    HAVE_LMFIT = False

_SAFE_FUNCS = {k: getattr(math, k) for k in ("exp", "log", "log10", "sqrt", "sin", "cos", "tan", "asin", "acos", "atan",
                                                "sinh", "cosh", "tanh", "fabs", "floor", "ceil", "pi", "e")}
_SAFE_FUNCS.update(abs=abs, min=min, max=max, pow=pow)
_ALLOWED_NODES = (ast.Expression, ast.BinOp, ast.UnaryOp, ast.Constant, ast.Name, ast.Load, ast.Call, ast.Add, ast.Sub, ast.Mult,
                  ast.Div, ast.Pow, ast.USub, ast.UAdd, ast.Mod, ast.FloorDiv, ast.IfExp, ast.Compare, ast.Lt, ast.Gt, ast.LtE,
                  ast.GtE, ast.Eq, ast.NotEq, ast.BoolOp, ast.And, ast.Or)


def _compile_expr(expr):
    tree = ast.parse(expr.strip(), mode="eval")
    for node in ast.walk(tree):
        if not isinstance(node, _ALLOWED_NODES):
            raise ValueError("unsupported syntax in parameter expression %r: %s" % (expr, type(node).__name__))
        if isinstance(node, ast.Call) and not (isinstance(node.func, ast.Name) and node.func.id in _SAFE_FUNCS):
            raise ValueError("unsupported function in parameter expression %r" % expr)
    return compile(tree, "<expr>", "eval")


class Parameter:
    def __init__(self, name, value=None, vary=True, min=-np.inf, max=np.inf, expr=None, brute_step=None):
        self.name = name
        self._val = None if value is None else float(value)
        self.min = -np.inf if min is None else float(min)
        self.max = np.inf if max is None else float(max)
        self.expr = expr
        self.vary = bool(vary) if expr is None else False
        self.brute_step = brute_step
        self.stderr = None
        self.correl = None
        self.init_value = self._val
        self._code = _compile_expr(expr) if expr is not None else None
        if self._val is not None and expr is None:
            self._val = float(np.clip(self._val, self.min, self.max))

    @property
    def value(self):
        return self._val

    @value.setter
    def value(self, v):
        self._val = float(v)

    def set(self, value=None, vary=None, min=None, max=None, expr=None, brute_step=None):
        if min is not None:
            self.min = float(min)
        if max is not None:
            self.max = float(max)
        if value is not None:
            self._val = float(value)
            self.expr, self._code = None, None
        if vary is not None:
            self.vary = bool(vary)
        if expr is not None:
            self.expr, self._code, self.vary = (expr, _compile_expr(expr), False) if expr != "" else (None, None, self.vary)
        if brute_step is not None:
            self.brute_step = brute_step

    def __repr__(self):
        s = "<Parameter '%s', value=%s" % (self.name, self._val)
        if self.expr is not None:
            s += ", expr='%s'" % self.expr
        elif not self.vary:
            s += " (fixed)"
        return s + ", bounds=[%s:%s]>" % (self.min, self.max)

    def __float__(self):
        return float(self._val)


class Parameters(OrderedDict):
    """Ordered ``{name: Parameter}`` with constraint expressions."""

    def add(self, name, value=None, vary=True, min=-np.inf, max=np.inf, expr=None, brute_step=None):
        if isinstance(name, Parameter):
            self[name.name] = name
        else:
            self[name] = Parameter(name, value=value, vary=vary, min=min, max=max, expr=expr, brute_step=brute_step)
        self.update_constraints()

    def add_many(self, *parlist):
        for p in parlist:
            self.add(*p) if isinstance(p, (tuple, list)) else self.add(p)

    def valuesdict(self):
        return OrderedDict((k, p.value) for k, p in self.items())

    def update_constraints(self):
        """Re-evaluates every ``expr`` parameter (dependencies resolved by repeated passes)."""
        pending = [p for p in self.values() if p._code is not None]
        for _ in range(len(pending) + 1):
            if not pending:
                return
            env = dict(_SAFE_FUNCS)
            env.update({k: p._val for k, p in self.items() if p._val is not None})
            nxt = []
            for p in pending:
                try:
                    p._val = float(eval(p._code, {"__builtins__": {}}, env))
                    env[p.name] = p._val
                except NameError:
                    nxt.append(p)
            if len(nxt) == len(pending):
                # unresolved names may be parameters that have not been added yet (lmfit defers too)
                return
            pending = nxt

    def copy(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = Parameters()
        for k, p in self.items():
            q = Parameter(k, value=p._val if p.expr is None else None, vary=p.vary, min=p.min, max=p.max, expr=p.expr,
                          brute_step=p.brute_step)
            q._val = p._val
            q.stderr, q.init_value = p.stderr, p.init_value
            OrderedDict.__setitem__(out, k, q)
        return out

    def pretty_print(self):
        for k, p in self.items():
            print("%-16s %-14.6g min=%-10.4g max=%-10.4g vary=%-5s expr=%s" % (k, p.value, p.min, p.max, p.vary, p.expr))


# ---- bounds transforms (MINUIT / lmfit.Parameter.setup_bounds, from_internal) ---------------------------------
def _to_internal(p):
    v, lo, hi = p.value, p.min, p.max
    if np.isfinite(lo) and np.isfinite(hi):
        return math.asin(min(1.0, max(-1.0, 2.0 * (v - lo) / (hi - lo) - 1.0)))
    if np.isfinite(lo):
        return math.sqrt(max((v - lo + 1.0) ** 2 - 1.0, 0.0))
    if np.isfinite(hi):
        return math.sqrt(max((hi - v + 1.0) ** 2 - 1.0, 0.0))
    return v


def _from_internal(p, x):
    lo, hi = p.min, p.max
    if np.isfinite(lo) and np.isfinite(hi):
        return lo + (math.sin(x) + 1.0) * (hi - lo) / 2.0
    if np.isfinite(lo):
        return lo - 1.0 + math.sqrt(x * x + 1.0)
    if np.isfinite(hi):
        return hi + 1.0 - math.sqrt(x * x + 1.0)
    return x


class MinimizerResult:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "<MinimizerResult method=%s nfev=%s success=%s residual=%s>" % (self.method, self.nfev, self.success, self.residual)


_SCALAR_METHODS = {"bfgs": "BFGS", "powell": "Powell", "nelder": "Nelder-Mead", "nelder-mead": "Nelder-Mead", "lbfgsb": "L-BFGS-B",
                   "l-bfgs-b": "L-BFGS-B", "cg": "CG", "cobyla": "COBYLA", "tnc": "TNC", "slsqp": "SLSQP",
                   "trust-constr": "trust-constr"}


def _dvalue_dinternal(p, x):
    """Derivative of ``_from_internal`` (the MINUIT bounds transform) with respect to the internal variable."""
    lo, hi = p.min, p.max
    if np.isfinite(lo) and np.isfinite(hi):
        return math.cos(x) * (hi - lo) / 2.0
    if np.isfinite(lo):
        return x / math.sqrt(x * x + 1.0)
    if np.isfinite(hi):
        return -x / math.sqrt(x * x + 1.0)
    return 1.0


_GRADIENT_METHODS = ("bfgs", "lbfgsb", "l-bfgs-b", "cg", "tnc", "slsqp", "trust-constr")


def _own_minimize(fcn, params, method="leastsq", args=None, kws=None, iter_cb=None, nan_policy="raise", max_nfev=None, fcn_grad=None,
                  **fit_kws):
    """``fcn_grad`` (extension, optional): callable ``(params, names, *args, **kws) -> (value, d value / d params[names].value)``.
    With a gradient-based method the optimiser then gets the analytic gradient (chain rule through the bounds transform applied
    here) instead of differencing ``fcn`` numerically: one call per iteration instead of nvar + 1."""
    from scipy.optimize import minimize as sp_minimize

    args = tuple(args) if args is not None else ()
    kws = dict(kws) if kws else {}
    work = copy.deepcopy(params)
    work.update_constraints()
    names = [k for k, p in work.items() if p.vary and p.expr is None]
    if not names:
        raise ValueError("no parameter varies")
    m = str(method).lower()
    if m in ("leastsq", "least_squares"):
        m = "bfgs"  # the objective here is a scalar; lmfit itself would reject least-squares methods for it
    if m not in _SCALAR_METHODS:
        raise ValueError("unsupported method %r (lmfit is not installed; available: %s)" % (method, sorted(_SCALAR_METHODS)))
    state = {"nfev": 0, "last": None}

    def objective(x):
        for k, xi in zip(names, x):
            work[k]._val = _from_internal(work[k], float(xi))
        work.update_constraints()
        r = fcn(work, *args, **kws)
        state["nfev"] += 1
        r = np.asarray(r, dtype=float)
        val = float((r * r).sum()) if r.size > 1 else float(r.reshape(-1)[0])
        if np.isnan(val) and nan_policy == "raise":
            raise ValueError("The model function generated NaN values and the fit aborted")
        state["last"] = val
        if iter_cb is not None:
            iter_cb(work, state["nfev"], val, *args, **kws)
        return val

    state["ngev"] = 0

    def objective_with_grad(x):
        for k, xi in zip(names, x):
            work[k]._val = _from_internal(work[k], float(xi))
        work.update_constraints()
        val, g = fcn_grad(work, names, *args, **kws)
        state["nfev"] += 1
        state["ngev"] += 1
        val = float(val)
        if np.isnan(val) and nan_policy == "raise":
            raise ValueError("The model function generated NaN values and the fit aborted")
        state["last"] = val
        if iter_cb is not None:
            iter_cb(work, state["nfev"], val, *args, **kws)
        gi = np.array([gv * _dvalue_dinternal(work[k], float(xi)) for gv, k, xi in zip(np.asarray(g, float), names, x)])
        if not np.isfinite(val):
            gi = np.zeros(len(names))
        return val, gi

    x0 = np.array([_to_internal(work[k]) for k in names], float)
    if fcn_grad is not None and m in _GRADIENT_METHODS:
        # A free parameter that starts exactly ON a bound (generate_params' default D0 = 0) sits where the derivative of the bounds transform
        # is zero: its exact internal gradient vanishes and a gradient method would never move it - finite differences leave the bound only
        # through their second-order term.  Start 1e-6 of the range (1e-6 for one-sided bounds) inside: d value / d internal = 1e-3 x range.
        for i, k in enumerate(names):
            pk = work[k]
            lo_side = np.isfinite(pk.min) and pk._val <= pk.min
            hi_side = np.isfinite(pk.max) and pk._val >= pk.max
            if lo_side or hi_side:
                rng_ = (pk.max - pk.min) if (np.isfinite(pk.min) and np.isfinite(pk.max)) else 1.0
                pk._val = (pk.min + 1e-6 * rng_) if lo_side else (pk.max - 1e-6 * rng_)
                x0[i] = _to_internal(pk)
    opts = dict(fit_kws.pop("options", {}))
    if max_nfev is not None:
        opts.setdefault("maxfev" if _SCALAR_METHODS[m] in ("Powell", "Nelder-Mead") else "maxiter", int(max_nfev))
    if fcn_grad is not None and m in _GRADIENT_METHODS:
        if m == "bfgs" and "gtol" not in opts:
            # With an exact gradient the default gtol (1e-5, absolute) is far below what a sum over 1e4..1e7 tracks can resolve:
            # the optimiser would spend dozens of evaluations in a line search that cannot improve a converged objective and
            # stop on "precision loss".  Scale it with the objective: a gradient of 1e-8 |f| moves f by less than its rounding.
            f0, _ = objective_with_grad(x0)
            if np.isfinite(f0):
                opts["gtol"] = max(1e-5, 1e-8 * abs(f0))
        res = sp_minimize(objective_with_grad, x0, method=_SCALAR_METHODS[m], jac=True, options=opts or None, **fit_kws)
    else:
        res = sp_minimize(objective, x0, method=_SCALAR_METHODS[m], options=opts or None, **fit_kws)
    final = objective(res.x)  # leaves `work` at the optimum
    for k in names:
        work[k].init_value = params[k].value
    return MinimizerResult(params=work, residual=np.atleast_1d(np.float64(final)), nfev=state["nfev"], ngev=state["ngev"], success=bool(res.success),
                           message=str(res.message), method=m, nvarys=len(names), var_names=names,
                           init_vals=[params[k].value for k in names], chisqr=float(final) ** 2, x=res.x, scipy_result=res,
                           aborted=False, errorbars=False)


def _to_own_parameters(params):
    """Our Parameters from any lmfit-style mapping {name: object with value / vary / min / max / expr}."""
    own = _OwnParameters()
    for k, p in params.items():
        expr = getattr(p, "expr", None) or None
        q = _OwnParameter(k, value=None if expr else p.value, vary=p.vary, min=p.min, max=p.max, expr=expr,
                          brute_step=getattr(p, "brute_step", None))
        q._val = float(p.value) if p.value is not None else None
        OrderedDict.__setitem__(own, k, q)
    own.update_constraints()
    return own


def minimize_with_gradient(fcn, params, method="bfgs", args=None, kws=None, nan_policy="propagate", fcn_grad=None, **fit_kws):
    """``minimize`` with an analytic gradient: always the built-in scipy driver (``_own_minimize``: MINUIT bounds transform + its chain
    rule).  With real lmfit installed the fit runs on a converted copy of the lmfit Parameters - routing a ``jac`` through lmfit's
    scalar minimisers would need its internal variable transform, which is unpinned here - and the result's ``.params`` is a copy of
    the caller's lmfit Parameters carrying the fitted values (``.value`` of every free and every ``expr`` parameter)."""
    own = params if isinstance(params, _OwnParameters) else _to_own_parameters(params)
    res = _own_minimize(fcn, own, method=method, args=args, kws=kws, nan_policy=nan_policy, fcn_grad=fcn_grad, **fit_kws)
    if not isinstance(params, _OwnParameters):
        out = copy.deepcopy(params)
        for k, p in res.params.items():
            if not getattr(out[k], "expr", None):
                out[k].value = p.value
        for k, p in res.params.items():  # expr parameters follow from the free ones; make sure the copy shows the same numbers
            if getattr(out[k], "expr", None) and hasattr(out, "update_constraints"):
                out.update_constraints()
                break
        res.own_params, res.params = res.params, out
    return res


_OwnParameters, _OwnParameter = Parameters, Parameter

if HAVE_LMFIT:  # pragma: no cover
    Parameters = _lmfit.Parameters  # noqa: F811
    Parameter = _lmfit.Parameter    # noqa: F811
    minimize = _lmfit.minimize
else:
    minimize = _own_minimize


def is_parameters(obj):
    return isinstance(obj, Parameters)
