#!/usr/bin/env python3
"""Times one objective evaluation / posterior annotation for the non-headline BASELINE configs (C3, C5) on the GPU."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1


def params(vals):
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


def timeit(f, n=3):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t0) / n, r


out = {}
# ---- C3: 3 states, lengths 5..50 geometric, F=6
S = 3
Ds = [0.0, 0.04, 0.25]
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
Fs = [0.3, 0.3, 0.4]
sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
tracks = {str(L): synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=L) for L, n in sizes.items() if n > 0}
vals = dict(D0=1e-4, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.07, p02=0.03, p10=0.05, p12=0.05, p20=0.03, p21=0.07, pBL=0.1)
_, lst, _ = T.engine.sort_buckets(tracks)
for F in (4, 6):
    ts = T.TrackSet(lst)
    model = T._objective_model(params(vals), ts, 0.02, [1], None, 3, 1, F, 1)
    dt, v = timeit(lambda: ts.loglik(model))
    out["C3_F%d" % F] = dict(tracks=ts.n_tracks, buckets=len(lst), s_per_eval=dt, ll=v, kernel_ms=ts.ctx.last_kernel_ms(), launch=ts.ctx.last_launch_info())
    ts.close()
# ---- C5: 4 states, ns=3, F=4, L=60
S = 4
Ds = [0.0, 0.02, 0.1, 0.5]
Tm = np.full((4, 4), 0.05 / 3); Tm[np.arange(4), np.arange(4)] = 0.95
Fs = [0.25] * 4
N = int(5e5 * scale)
Cs = synth.brownian_tracks(N, 60, Ds, Tm, Fs, seed=2)
vals = dict(D0=1e-4, D1=0.02, D2=0.1, D3=0.5, LocErr=0.02, F0=.25, F1=.25, F2=.25, F3=.25, pBL=0.1)
for i in range(4):
    for j in range(4):
        if i != j:
            vals["p%d%d" % (i, j)] = 0.05 / 3
ts = T.TrackSet([Cs])
model = T._objective_model(params(vals), ts, 0.02, [1], None, 4, 3, 4, 1)
dt, v = timeit(lambda: ts.loglik(model), n=2)
out["C5_LL_ns3_F4"] = dict(tracks=N, s_per_eval=dt, ll=v, kernel_ms=ts.ctx.last_kernel_ms(), launch=ts.ctx.last_launch_info())
model = T._objective_model(params(vals), ts, 0.02, [1], None, 4, 1, 5, 1)
dt, v = timeit(lambda: ts.loglik(model), n=2)
out["C5_LL_ns1_F5"] = dict(tracks=N, s_per_eval=dt, kernel_ms=ts.ctx.last_kernel_ms(), launch=ts.ctx.last_launch_info())
t0 = time.perf_counter()
pr = ts.predict(model)
out["C5_predict_ns1_F5"] = dict(tracks=N, s=time.perf_counter() - t0, kernel_ms=ts.ctx.last_kernel_ms(), launch=ts.ctx.last_launch_info(),
                                rowsum_err=float(np.abs(pr[0].sum(-1) - 1).max()))
ts.close()
# ---- 2-state posteriors at C2 size
Cs = synth.brownian_tracks(int(1e6 * scale), 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
vals = dict(D0=0.0, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
ts = T.TrackSet([Cs])
model = T._objective_model(params(vals), ts, 0.02, [1], None, 2, 1, 6, 1)
t0 = time.perf_counter()
pr = ts.predict(model)
out["C2_predict_F6"] = dict(tracks=len(Cs), s=time.perf_counter() - t0, kernel_ms=ts.ctx.last_kernel_ms(), launch=ts.ctx.last_launch_info())
ts.close()
for k, v in out.items():
    print(k, json.dumps(v))
