#!/usr/bin/env python3
"""Traces the C1 fit with the analytic gradient: objective / gradient norm per evaluation (where do the evaluations go?)."""
import sys, os, json, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import tracking as T, lmfit_compat as LC
info = json.load(open("tests/golden/c1_simfov_10k.json"))
data = np.load("tests/golden/c1_simfov_10k.npz")
tr = {k: data["tr_" + k] for k in info["keys"]}
p0 = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1], estimated_Fs=[0.5],
                       estimated_transition_rates=0.05)
orig = T.cum_Proba_Cs_grad
log = []
def wrapped(params, names, *a, **k):
    v, g = orig(params, names, *a, **k)
    log.append((v, float(np.abs(g).max())))
    return v, g
T.cum_Proba_Cs_grad = wrapped
for opts in (None, dict(gtol=1e-3), dict(gtol=1e-2)):
    log.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        kw = {} if opts is None else dict(options=opts)
        _, tracks, sig = T.engine.sort_buckets(tr)
        ts = T.TrackSet(tracks)
        fit = LC.minimize(T.cum_Proba_Cs, p0, args=(ts, info["dt"], [1], None, 2, 1, 6, 0, 1, 1, 0.2, 120, 2000, None, "window"), method="bfgs",
                          nan_policy="propagate", fcn_grad=wrapped, **kw)
        ts.close()
    print(opts, "nfev", fit.nfev, "nit", fit.scipy_result.nit, fit.message, fit.residual[0])
    for i, (v, g) in enumerate(log):
        if i % 4 == 0 or i > len(log) - 25:
            print("   %3d %.9f %.3e" % (i, v, g))
