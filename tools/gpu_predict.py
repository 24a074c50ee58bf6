#!/usr/bin/env python3
"""Posterior (predict) kernel timing on one bucket: tools/gpu_predict.py S F N L"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
S, F, N, L = [int(x) for x in sys.argv[1:5]]
Ds = [0.0, 0.05, 0.25, 0.8][:S]
Tm = np.full((S, S), 0.05); Tm[np.arange(S), np.arange(S)] = 1 - 0.05 * (S - 1)
Cs = synth.brownian_tracks(N, L, Ds, Tm, [1.0 / S] * S, seed=1)
vals = {"LocErr": 0.02, "pBL": 0.1}
for i in range(S):
    vals["D%d" % i] = Ds[i] + 1e-4 * (i == 0)
    vals["F%d" % i] = 1.0 / S
    for j in range(S):
        if i != j:
            vals["p%d%d" % (i, j)] = 0.05
p = Parameters()
for k, v in vals.items():
    p.add(k, value=v)
ts = T.TrackSet([Cs])
model = T._objective_model(p, ts, 0.02, [1], None, S, 1, F, 1)
ts.loglik(model); ll_ms = []
for _ in range(3):
    ts.loglik(model); ll_ms.append(ts.ctx.last_kernel_ms())
ms = []
for _ in range(3):
    t0 = time.perf_counter(); pr = ts.predict(model); wall = time.perf_counter() - t0
    ms.append(ts.ctx.last_kernel_ms())
print(json.dumps(dict(S=S, F=F, N=N, L=L, predict_kernel_ms=float(np.median(ms)), ll_kernel_ms=float(np.median(ll_ms)), predict_wall_s=wall, launch=ts.ctx.last_launch_info())))
