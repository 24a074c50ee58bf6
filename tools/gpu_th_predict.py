"""Timing of the threshold-fusion posteriors (predict_Bs, nb_max=1) on the GPU box."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

for name, N, L, Ds, Tm, Fs, F, thr, mx in [
        ("C2-size", 200_000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], 5, 0.1, 200),
        ("C5-size", 50_000, 60, [0.0, 0.02, 0.1, 0.5], [[.85, .05, .05, .05], [.05, .85, .05, .05], [.05, .05, .85, .05], [.05, .05, .05, .85]], [.25] * 4, 5, 0.1, 200)]:
    X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
    ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
    ts = TrackSet([X])
    model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, F)
    t0 = time.perf_counter()
    pr = ts.predict_th(model, thr, mx, 1)[0]
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    pr = ts.predict_th(model, thr, mx, 1)[0]
    wall2 = time.perf_counter() - t0
    print("%s: N=%d L=%d S=%d: predict_th wall %.1f ms / %.1f ms (kernel %.1f ms) -> %.0f tracks/s, sum check %.3e"
          % (name, N, L, len(Ds), wall * 1e3, wall2 * 1e3, ts.ctx.last_kernel_ms(), N / wall2, np.abs(pr.sum(-1) - 1).max()), flush=True)
    ts.close()
