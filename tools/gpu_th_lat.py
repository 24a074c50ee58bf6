#!/usr/bin/env python3
"""Threshold-fusion evaluation latency: C2 (1e6 x 30, one bucket) and a C1-like small dataset (6 730 tracks in 16 buckets)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
p = Parameters()
for k, v in vals.items():
    p.add(k, value=v)
Ds2, Tm2, Fs2 = [0.0, 0.25], [[0.9, 0.1], [0.1, 0.9]], [0.6, 0.4]
sizes = synth.bucket_sizes_geometric(6730, list(range(5, 21)), 0.85)
small = {str(L): synth.brownian_tracks(n, L, Ds2, Tm2, Fs2, seed=L) for L, n in sizes.items() if n > 0}
big = {"30": synth.brownian_tracks(1000000, 30, Ds2, Tm2, Fs2, seed=0)}
for name, tr, n in (("C1-like", small, 200), ("C2", big, 30)):
    _, lst, _ = T.engine.sort_buckets(tr)
    ts = T.TrackSet(lst)
    model = T._objective_model(p, ts, 0.02, [1.0], None, 2, 1, 6, 1)
    for _ in range(4):
        v = ts.loglik_th(model, 0.2, 120, 2000)
    t0 = time.perf_counter()
    ks = []
    for _ in range(n):
        v = ts.loglik_th(model, 0.2, 120, 2000)
        ks.append(ts.ctx.last_kernel_ms())
    dt = (time.perf_counter() - t0) / n
    for _ in range(4):
        w = ts.loglik(model)
    t0 = time.perf_counter()
    for _ in range(n):
        w = ts.loglik(model)
    dw = (time.perf_counter() - t0) / n
    print("%-8s threshold: %.3f ms/eval (kernels %.3f ms)   window: %.3f ms/eval   -LL_th %.6f" % (name, dt * 1e3, np.mean(ks), dw * 1e3, -v), flush=True)
    ts.close()
