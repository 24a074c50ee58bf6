"""Threshold-fusion evaluation at the C5 model (4 states, nb_substeps 3, len 60) on the GPU box: feasibility + timing."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

Ds = [0.0, 0.02, 0.1, 0.5]
Tm = np.array([[.85, .05, .05, .05], [.05, .85, .05, .05], [.05, .05, .85, .05], [.05, .05, .05, .85]])
Fs = np.array([.25] * 4)
ns = 3
Tsub = 1 - np.exp(-(Tm - np.diag(np.diag(Tm))) / ns)
Tsub[np.arange(4), np.arange(4)] = 0
Tsub[np.arange(4), np.arange(4)] = 1 - Tsub.sum(1)
ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
for N in (20_000, 500_000):
    X = synth.brownian_tracks(N, 60, Ds, Tm, Fs, seed=2)
    ts = TrackSet([X])
    model = ts.make_model(np.array([[[0.02]]]), ds, Fs, Tsub, 0.1, (1.0,), ns, 4)
    t0 = time.perf_counter()
    tot = ts.loglik_th(model, 0.2, 120, 2000)
    w1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    tot = ts.loglik_th(model, 0.2, 120, 2000)
    w2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    fw = ts.loglik(model)
    w3 = time.perf_counter() - t0
    print("C5 ns=3 N=%d: th total %.4f wall %.1f / %.1f ms | fixed-window (F=4) total %.4f %.1f ms | launch %s"
          % (N, tot, w1 * 1e3, w2 * 1e3, fw, w3 * 1e3, ts.ctx.last_launch_info()), flush=True)
    ts.close()
