"""Sweep of the threshold-fusion apply-kernel geometry knobs on the GPU box (C2-size input)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

N, L = 1_000_000, 30
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
for tt, thr, ov, sg in [(0, 0, 2, 0), (64, 768, 2, 1), (64, 512, 2, 1), (64, 1024, 2, 1), (64, 256, 2, 1), (64, 768, 4, 1), (64, 768, 2, 0), (64, 768, 4, 0), (64, 768, 8, 0)]:
    for k, v in (("EXTRACK_TH_TT", tt), ("EXTRACK_TH_THREADS", thr), ("EXTRACK_TH_OVERSUB", ov), ("EXTRACK_TH_SINGLE", sg)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    ts = TrackSet([X])
    model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, 6)
    ts.loglik_th(model, 0.2, 120, 2000)
    t0 = time.perf_counter()
    for _ in range(5):
        tot = ts.loglik_th(model, 0.2, 120, 2000)
    wall = (time.perf_counter() - t0) / 5
    print("single=%d " % sg, end=""); print("TT=%d threads=%d oversub=%d: wall %.2f ms total %.4f launch %s" % (tt, thr, ov, wall * 1e3, tot, ts.ctx.last_launch_info()), flush=True)
    ts.close()
