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
    _CSAFE.pop(_k, None)  # not analytic: an expression using them falls back to finite differences


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
    pending = [p for p in params.values() if getattr(p, "_code", None) is not None]
    for _ in range(len(pending) + 1):
        if not pending:
            break
        env = dict(_CSAFE)
        env.update(vals)
        nxt = []
        for p in pending:
            try:
                vals[p.name] = complex(eval(p._code, {"__builtins__": {}}, env))
            except NameError:
                nxt.append(p)
        if len(nxt) == len(pending):
            raise NameError("unresolved names in parameter expressions")
        pending = nxt
    return vals


def model_tangents(params, dt, nb_substeps, Matrix_type, cell_dims, names, has_sigma=False):
    """Tangent dicts (the ``tangents`` argument of ``_lib.Context.loglik_grad``), one per free parameter in ``names``."""
    from .tracking import _extract_arrays
    base = {k: _V(p.value) for k, p in params.items()}
    _, Ds0, _, _, _, _ = _extract_arrays(base, dt, nb_substeps, Matrix_type)
    S = len(Ds0)
    ds = np.sqrt(2 * np.real(Ds0) * dt)
    _, dps = engine.p_stay_table_grad(ds, S, nb_substeps, cell_dims)  # [G], [G, S]
    out = []
    for n in names:
        cv = values_along(params, n)
        le, Ds, Fs, TrMat, pBL, so = _extract_arrays({k: _V(v) for k, v in cv.items()}, dt, nb_substeps, Matrix_type)
        t = dict(ds2=2 * dt * np.imag(Ds) / _H, Fs=np.imag(Fs) / _H, TrMat=np.imag(TrMat) / _H, pBL=float(np.imag(pBL) / _H))
        t["p_stay"] = dps @ t["ds2"]
        if has_sigma:
            if so is not None:
                t["slope"], t["offset"] = float(np.imag(so[0]) / _H), float(np.imag(so[1]) / _H)
        else:
            t["locerr"] = np.imag(le) / _H
        out.append(t)
    return out


def objective_and_gradient(params, ts, dt, cell_dims, nb_states, nb_substeps, frame_len, Matrix_type=1, comm=None, names=None):
    """(-sum LL, d(-sum LL)/d(free parameter VALUES)) at ``params`` on the TrackSet ``ts``; (+inf, zeros) for invalid parameters.
    ``comm``: extrack_amd.distributed.Comm - the (1 + nvar) vector is all-reduced over the ranks."""
    from .tracking import _objective_model
    names = free_names(params) if names is None else list(names)
    model = _objective_model(params, ts, dt, cell_dims, None, nb_states, nb_substeps, frame_len, Matrix_type)
    if model is None:
        return np.inf, np.zeros(len(names))
    tang = model_tangents(params, dt, nb_substeps, Matrix_type, cell_dims, names, has_sigma=ts.has_sigma)
    if ts.n_tracks:
        ll, g = ts.ctx.loglik_grad(model, tang)
    else:
        ll, g = 0.0, np.zeros(len(names))
    if comm is not None:
        v = comm.allreduce_vector(np.concatenate([[ll], g]))
        ll, g = float(v[0]), v[1:]
    return -ll, -np.asarray(g)
