"""Wall and kernel time per log-likelihood evaluation of an N x 30 2-state bucket through the C ABI only (numpy + ctypes, no torch):
python tools/gpu_small_ll.py [n_tracks] [n_evals]   (EXTRACK_OVERSUB etc. from the environment)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
Cs = synth.brownian_tracks(n, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1)
ts = T.TrackSet([Cs])
p = T.Parameters()
for k, v in dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1).items():
    p.add(k, value=v)
model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
for _ in range(100):
    ts.loglik(model)
ks = []
t0 = time.perf_counter()
for _ in range(reps):
    v = ts.loglik(model)
    ks.append(ts.ctx.last_kernel_ms())
dt = (time.perf_counter() - t0) / reps * 1e3
print("oversub=%s n=%d: %.4f ms/eval, kernel %.4f ms (min %.4f), host %.1f us, launch %s, value %.6f"
      % (os.environ.get("EXTRACK_OVERSUB", "default"), n, dt, np.mean(ks), np.min(ks), (dt - np.mean(ks)) * 1e3, ts.ctx.last_launch_info(), v), flush=True)
