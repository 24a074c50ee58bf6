"""Plan-kernel block size sweep (threshold-fusion evaluation, C2 / 3-state / 4-state inputs) on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

CASES = [("C2", 1_000_000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], 6, 0.2, 120),
         ("S3", 1_000_000, 30, [0.0, 0.04, 0.25], [[.9, .05, .05], [.05, .9, .05], [.05, .05, .9]], [.3, .3, .4], 6, 0.2, 120),
         ("S4", 500_000, 60, [0.0, 0.02, 0.1, 0.5], [[.85, .05, .05, .05], [.05, .85, .05, .05], [.05, .05, .85, .05], [.05, .05, .05, .85]],
          [.25] * 4, 5, 0.1, 200)]
for name, N, L, Ds, Tm, Fs, F, thr, mx in CASES:
    X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
    ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
    for pt, pl in ((512, 4), (512, 30), (256, 30), (1024, 30)):
        os.environ["EXTRACK_TH_PLAN_THREADS"] = str(pt)
        os.environ["EXTRACK_TH_PAIR_LANES"] = str(pl)
        ts = TrackSet([X])
        model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, F)
        for _ in range(2):
            ts.loglik_th(model, thr, mx, 2000)
        t0 = time.perf_counter()
        for _ in range(3):
            tot = ts.loglik_th(model, thr, mx, 2000)
        wall = (time.perf_counter() - t0) / 3
        print("pair_lanes<=%d " % pl, end=""); print("%s plan_threads=%d: wall %.2f ms total %.4f" % (name, pt, wall * 1e3, tot), flush=True)
        ts.close()
