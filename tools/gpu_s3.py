#!/usr/bin/env python3
"""Single-bucket 3-state workload (S=3, F=6, L=30) for profiling the general kernel."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 6
Ds = [0.0, 0.04, 0.25]
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
Cs = synth.brownian_tracks(N, 30, Ds, Tm, [0.3, 0.3, 0.4], seed=1)
vals = dict(D0=1e-4, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.07, p02=0.03, p10=0.05, p12=0.05, p20=0.03, p21=0.07, pBL=0.1)
p = Parameters()
for k, v in vals.items():
    p.add(k, value=v)
ts = T.TrackSet([Cs])
model = T._objective_model(p, ts, 0.02, [1], None, 3, 1, F, 1)
ts.loglik(model)
ms = []
for _ in range(5):
    v = ts.loglik(model)
    ms.append(ts.ctx.last_kernel_ms())
print(json.dumps(dict(N=N, F=F, kernel_ms=float(np.mean(ms)), ll=v, launch=ts.ctx.last_launch_info(),
                      ps_per_seq_step=float(np.mean(ms)) * 1e-3 / (N * 29 * 3 ** F) * 1e12)))
