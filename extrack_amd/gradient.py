"""Analytic gradient of the fit objective: host side of ``extrack_loglik_grad`` (include/extrack_hip.h).

The reference fits with ``lmfit.minimize(cum_Proba_Cs, ...)`` (extrack/tracking.py:1371): the optimiser differentiates the objective
by finite differences, nvar + 1 evaluations per BFGS iteration.  Here ONE launch returns -sum(LL) and its exact gradient with
respect to the free fit parameters:

    free parameter values  --(expr constraints)-->  all parameter values  --(extract_params, tracking.py:913-986)-->
    model arrays (ds^2, Fs, TrMat, LocErr, slope/offset, pBL)  --(FOV table, tracking.py:182-191)-->  p_stay
    --(HIP kernel, forward-mode tangents inside the recursion)-->  d sum(LL) / d theta

The first two maps are differentiated by the complex-step method (exact to rounding: they are compositions of analytic
functions - products, exp, the matrix exponential), the p_stay table analytically, the recursion on the GPU.  The bounds
transform of the optimiser's internal variables is applied in ``lmfit_compat``.
"""
import cmath

import numpy as np

from . import engine
from .lmfit_compat import _SAFE_FUNCS

_H = 1e-30
_CSAFE = dict(_SAFE_FUNCS)
_CSAFE.update({k: getattr(cmath, k) for k in ("exp", "log", "log10", "sqrt", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh",
                                                "tanh")})
for _k in ("fabs", "floor", "ceil", "abs", "min", "max"):
    _CSAFE.pop(_k, None)  # not analytic: param_fitting then differentiates the objective by finite differences (analytic_support)


class _V:
    __slots__ = ("value",)

    def __init__(self, v):
        self.value = v


def free_names(params):
    return [k for k, p in params.items() if p.vary and p.expr is None]


def values_along(params, name, h=_H):
    """{param name: complex value} with the free parameter ``name`` displaced by i*h and every ``expr`` parameter re-evaluated in
    complex arithmetic.  Raises TypeError / NameError when an expression is not complex-differentiable."""
    vals = {k: complex(p.value) for k, p in params.items()}
    vals[name] += 1j * h
    exprs = [p for p in params.values() if getattr(p, "_code", None) is not None]
    env = dict(_CSAFE)
    env.update(vals)
    # expressions may depend on each other in any order (c = 'b*3', b = '2*a'): sweep until nothing changes (a chain of depth d settles in d sweeps)
    for _ in range(len(exprs) + 1):
        changed = False
        for p in exprs:
            new = complex(eval(p._code, {"__builtins__": {}}, env))
            changed = changed or new != vals[p.name]
            vals[p.name] = env[p.name] = new
        if not changed:
            break
    return vals


_NSAFE = {"exp": np.exp, "log": np.log, "log10": np.log10, "sqrt": np.sqrt, "sin": np.sin, "cos": np.cos, "tan": np.tan, "asin": np.arcsin,
          "acos": np.arccos, "atan": np.arctan, "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh, "pi": np.pi, "e": np.e, "pow": np.power}
_CODE_CACHE = {}


def _expr_code(p):
    """Compiled constraint expression of a parameter (ours carry it; an lmfit Parameter only has the string)."""
    code = getattr(p, "_code", None)
    if code is None and getattr(p, "expr", None):
        code = _CODE_CACHE.get(p.expr)
        if code is None:
            from .lmfit_compat import _compile_expr
            code = _CODE_CACHE[p.expr] = _compile_expr(p.expr)
    return code


def _values_batched(params, names, h=_H):
    """{param name: complex array [n_dir]}: direction i displaces the free parameter names[i] by i*h; every ``expr`` parameter is
    re-evaluated on the whole batch at once (numpy complex arithmetic).  Raises TypeError / NameError / ValueError when an
    expression is not complex-differentiable (abs, min, max, comparisons, ...)."""
    n = len(names)
    vals = {}
    for k, p in params.items():
        vals[k] = np.full(n, complex(p.value))
    for i, nm in enumerate(names):
        vals[nm][i] += 1j * h
    exprs = [(k, _expr_code(p)) for k, p in params.items() if getattr(p, "expr", None)]
    env = dict(_NSAFE)
    env.update(vals)
    # expressions may depend on each other in any order (c = 'b*3', b = '2*a'): every sweep re-evaluates all of them against the values
    # AND tangents refreshed so far, until nothing changes (a chain of depth d settles in d sweeps)
    for _ in range(len(exprs) + 1):
        changed = False
        for k, code in exprs:
            try:
                new = np.asarray(eval(code, {"__builtins__": {}}, env), complex) * np.ones(n)
            except NameError as e:
                if any(nm in str(e) for nm in ("abs", "min", "max", "fabs", "floor", "ceil")):
                    raise TypeError("parameter expression is not analytic: %s" % e)
                raise
            changed = changed or not np.array_equal(new, vals[k])
            vals[k] = env[k] = new
        if not changed:
            break
    return vals


def _extract_batched(vals, nb_substeps, Matrix_type):
    """``tracking._extract_arrays`` on a batch: vals {name: complex [n]} -> le [k, n], Ds [S, n], Fs [S, n], TrMat [S, S, n], pBL [n],
    (slope [n], offset [n]) or None.  Matrix types with a matrix exponential go direction by direction."""
    names = np.sort(list(vals.keys()))
    n = len(next(iter(vals.values())))
    le = np.array([vals[k] for k in names if k.startswith("LocErr")]).reshape(-1, n)
    so = (vals["slope_LocErr"], vals["offset_LocErr"]) if "slope_LocErr" in vals else None
    Ds = np.array([vals[k] for k in names if k.startswith("D") and len(k) < 3])
    Fs = np.array([vals[k] for k in names if k.startswith("F")])
    S = len(Ds)
    pBL = vals.get("pBL")
    if Matrix_type in (0, 1):
        T = np.zeros((S, S, n), complex)
        for k in vals:
            if k != "pBL" and k.startswith("p"):
                T[int(k[1]), int(k[2])] = vals[k]
        T = T / nb_substeps
        if Matrix_type == 1:
            T = 1 - np.exp(-T)
        d = np.arange(S)
        T[d, d] = 0
        T[d, d] = 1 - T.sum(1)
        return le, Ds, Fs, T, pBL, so
    from .tracking import _extract_arrays
    T = np.zeros((S, S, n), complex)
    for i in range(n):
        T[:, :, i] = _extract_arrays({k: _V(complex(v[i])) for k, v in vals.items()}, 1.0, nb_substeps, Matrix_type)[3]
    return le, Ds, Fs, T, pBL, so


def model_tangents(params, dt, nb_substeps, Matrix_type, cell_dims, names, has_sigma=False):
    """Tangents of the model arrays along the free parameters ``names``, packed for ``_lib.Context.loglik_grad``: dict of arrays with
    the direction as FIRST axis (ds2 [n, S], Fs [n, S], TrMat [n, S, S], p_stay [n, G], pBL [n], locerr [n, k] or slope / offset [n]).
    One batched complex-step evaluation of the parameter plumbing (expr constraints -> extract_params) for all directions."""
    vals = _values_batched(params, names)
    le, Ds, Fs, TrMat, pBL, so = _extract_batched(vals, nb_substeps, Matrix_type)
    S = len(Ds)
    ds = np.sqrt(2 * np.real(Ds[:, 0]) * dt)
    _, dps = engine.p_stay_table_grad_cached(ds, S, nb_substeps, cell_dims)  # [G], [G, S]
    n = len(names)
    t = dict(ds2=np.ascontiguousarray(2 * dt * np.imag(Ds).T / _H), Fs=np.ascontiguousarray(np.imag(Fs).T / _H),
             TrMat=np.ascontiguousarray(np.moveaxis(np.imag(TrMat), 2, 0) / _H),
             pBL=(np.imag(pBL) / _H if pBL is not None else np.zeros(n)))
    t["p_stay"] = np.ascontiguousarray(t["ds2"] @ dps.T)
    if has_sigma:
        if so is not None:
            t["slope"], t["offset"] = np.imag(so[0]) / _H, np.imag(so[1]) / _H
    else:
        t["locerr"] = np.ascontiguousarray(np.imag(le).T / _H)
    return t


def tangent_rows(t):
    """The packed tangents of ``model_tangents`` as a list of per-direction dicts (tests, diagnostics)."""
    n = len(t["pBL"])
    return [{k: v[i] for k, v in t.items()} for i in range(n)]


def analytic_support(params, names):
    """None if every constraint expression is complex-differentiable along ``names``, else the reason (str): the caller then
    differentiates the objective by finite differences instead."""
    try:
        _values_batched(params, names)
    except (TypeError, NameError, ValueError, ZeroDivisionError, AttributeError) as e:
        return "%s: %s" % (type(e).__name__, e)
    return None


def objective_and_gradient(params, ts, dt, cell_dims, nb_states, nb_substeps, frame_len, Matrix_type=1, comm=None, names=None,
                           threshold_fusion=None):
    """(-sum LL, d(-sum LL)/d(free parameter VALUES)) at ``params`` on the TrackSet ``ts``; (+inf, zeros) for invalid parameters.
    ``comm``: extrack_amd.distributed.Comm - the (1 + nvar) vector is all-reduced over the ranks.
    ``threshold_fusion``: None = the fixed-window objective; (threshold, max_nb_states, chunk) = the threshold-fusion objective of
    extrack/tracking.py:427-743 and its gradient at the frozen plan of this evaluation (extrack_loglik_th_grad)."""
    from .tracking import _objective_model
    names = free_names(params) if names is None else list(names)
    if ts.has_dt and threshold_fusion is not None:
        raise NotImplementedError("the frozen-plan gradient does not serve per-track time steps: use gradient='fd'")
    model = _objective_model(params, ts, dt, cell_dims, None, nb_states, nb_substeps, frame_len, Matrix_type)
    if model is None:
        return np.inf, np.zeros(len(names))
    if not names:  # nothing to differentiate: the plain objective
        if threshold_fusion is not None:
            ll = comm.allreduce_loglik_th(ts, model, *threshold_fusion) if comm is not None else ts.loglik_th(model, *threshold_fusion)
        else:
            ll = comm.allreduce_loglik(ts, model) if comm is not None else ts.loglik(model)
        return -float(ll), np.zeros(0)
    tang = model_tangents(params, dt, nb_substeps, Matrix_type, cell_dims, names, has_sigma=ts.has_sigma)

    if threshold_fusion is not None:
        thr, mnb, chunk = threshold_fusion
        if comm is not None:
            v = comm.allreduce_loglik_th_grad(ts, model, tang, len(names), thr, mnb, chunk)
        elif ts.n_tracks:
            ll, g = ts.ctx.loglik_th_grad(model, tang, thr, mnb, chunk)
            v = np.concatenate([[ll], g])
        else:
            v = np.zeros(1 + len(names))
    elif comm is not None:
        v = comm.allreduce_loglik_grad(ts, model, tang, len(names))
    elif ts.n_tracks:
        ll, g = ts.ctx.loglik_grad(model, tang)
        v = np.concatenate([[ll], g])
    else:
        v = np.zeros(1 + len(names))
    ll, g = float(v[0]), v[1:]
    return -ll, -np.asarray(g)
