#!/usr/bin/env python3
"""BASELINE configs[2]: 1e6 tracks, 3 states, mixed lengths 5-50, full param_fitting (own lmfit-compatible BFGS) on one MI355X."""
import sys, os, time, json, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
F = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fusion = sys.argv[3] if len(sys.argv) > 3 else "window"  # "threshold": the kernel extrack.tracking.param_fitting runs in v1.6.3
grad = sys.argv[4] if len(sys.argv) > 4 else None             # "analytic" | "fd" | default: the timing probe decides
grad = None if grad in (None, "none", "default") else grad
Ds = [0.0, 0.04, 0.25]
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
Fs = [0.3, 0.3, 0.4]
sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
t0 = time.time()
tracks = {str(L): synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=L) for L, n in sizes.items() if n > 0}
print("data: %d tracks in %d buckets (%.1f s)" % (sum(len(v) for v in tracks.values()), len(tracks), time.time() - t0))
p0 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                       estimated_Fs=[0.33, 0.33], estimated_transition_rates=0.1)
buf = io.StringIO()
t0 = time.time()
with contextlib.redirect_stdout(buf):
    fit = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, nb_substeps=1, frame_len=F, verbose=0, method="bfgs", cell_dims=[1],
                          threshold=0.2, max_nb_states=120, fusion=fusion, gradient=grad)
dt = time.time() - t0
v = {k: round(fit.params[k].value, 5) for k in fit.params}
print(json.dumps(dict(fusion=fusion, frame_len=F, gradient=grad, gradient_path=getattr(fit, "gradient_path", None), gradient_why=getattr(fit, "gradient_why", None),
                      seconds=dt, nfev=fit.nfev, ngev=int(getattr(fit, "ngev", 0)), s_per_eval=dt / fit.nfev, success=fit.success, message=str(getattr(fit, "message", "")),
                      neg_ll=float(fit.residual[0]), params=v)))
# truth: D = 0, 0.04, 0.25; LocErr 0.02; F = .3 .3 .4; per-step transition probabilities Tm -> rates -ln(1 - p) for Matrix_type 1
print("true rates p01 %.4f p02 %.4f p10 %.4f p12 %.4f p20 %.4f p21 %.4f" % tuple(-np.log(1 - Tm[i, j]) for i, j in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1))))
