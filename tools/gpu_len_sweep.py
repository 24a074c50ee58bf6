#!/usr/bin/env python3
"""Kernel time of the 2-state fast path vs track length (fixed total positions): separates per-step from per-track cost."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
p = Parameters()
for k, v in dict(D0=0.0, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1).items():
    p.add(k, value=v)
rows = []
for L in (6, 10, 30, 60, 120, 31, 33):
    N = int(3e7 // L)
    Cs = synth.brownian_tracks(N, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=L)
    ts = T.TrackSet([Cs])
    model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
    ts.loglik(model)
    ms = []
    for _ in range(5):
        ts.loglik(model)
        ms.append(ts.ctx.last_kernel_ms())
    ts.close()
    rows.append((L, N, float(np.median(ms))))
    print(L, N, "%.3f ms" % rows[-1][2], "ns/track %.2f" % (rows[-1][2] * 1e6 / N), "ns/track-step %.4f" % (rows[-1][2] * 1e6 / N / (L - 1)))
# least squares: t = N * (a + b*(L-1))
A = np.array([[n, n * (l - 1)] for l, n, _ in rows[:5]], float)
y = np.array([t for _, _, t in rows[:5]])
a, b = np.linalg.lstsq(A, y, rcond=None)[0]
print("per-track fixed %.3f ns = %.1f steps; per step %.4f ns" % (a * 1e6, a / b, b * 1e6))
