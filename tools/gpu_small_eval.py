#!/usr/bin/env python3
"""Per-evaluation wall time on a small multi-bucket dataset (C1 fixture): where does the host overhead go?"""
import sys, os, time, json, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import tracking as T
from extrack_amd.lmfit_compat import Parameters
info = json.load(open("tests/golden/c1_simfov_10k.json"))
data = np.load("tests/golden/c1_simfov_10k.npz")
tr = {k: data["tr_" + k] for k in info["keys"]}
_, lst, _ = T.engine.sort_buckets(tr)
p = Parameters()
for k, v in info["values"].items():
    p.add(k, value=v)
ts = T.TrackSet(lst)
f = lambda: T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0)
import contextlib
with contextlib.redirect_stdout(io.StringIO()):
    f()
    t0 = time.perf_counter()
    for _ in range(300):
        v = f()
    dt = (time.perf_counter() - t0) / 300
    model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
    t0 = time.perf_counter()
    for _ in range(300):
        ts.loglik(model)
    dt2 = (time.perf_counter() - t0) / 300
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        f()
    pr.disable()
print("cum_Proba_Cs %.1f us/eval, C-ABI loglik only %.1f us, kernels %.1f us, buckets %d, value %.6f" % (dt * 1e6, dt2 * 1e6, ts.ctx.last_kernel_ms() * 1e3, len(lst), v))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(12); print(s.getvalue()[:2500])
