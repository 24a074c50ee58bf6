#!/usr/bin/env python3
"""BASELINE configs[3] size on ONE GPU (1e7 tracks x 30): index arithmetic / memory at scale, shard additivity."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
p = Parameters()
for k, v in dict(D0=0.0, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1).items():
    p.add(k, value=v)
t0 = time.time()
Cs = synth.brownian_tracks(N, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
print("generated %.1f GB in %.1f s" % (Cs.nbytes / 1e9, time.time() - t0))
t0 = time.time()
ts = T.TrackSet([Cs], min_len=30, max_len=30)
print("upload %.2f s" % (time.time() - t0))
model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
tot = ts.loglik(model)
ms = []
for _ in range(5):
    tot = ts.loglik(model)
    ms.append(ts.ctx.last_kernel_ms())
ts.close()
# first 1e6 tracks must reproduce the bench's value; eight 1/8 shards must add up to the whole
ts1 = T.TrackSet([Cs[:1000000]], min_len=30, max_len=30)
t1 = ts1.loglik(T._objective_model(p, ts1, 0.02, [1], None, 2, 1, 6, 1))
ts1.close()
parts = 0.0
for r in range(8):
    a, b = r * N // 8, (r + 1) * N // 8
    t8 = T.TrackSet([Cs[a:b]], min_len=30, max_len=30)
    parts += t8.loglik(T._objective_model(p, t8, 0.02, [1], None, 2, 1, 6, 1))
    t8.close()
print(json.dumps(dict(N=N, kernel_ms=float(np.median(ms)), evals_per_s_1e6_units=N / 1e6 / (np.median(ms) * 1e-3), total=tot, first_1e6=t1,
                      shard_sum_rel_err=abs(parts - tot) / abs(tot))))
