"""Diagnostics of the threshold-fusion evaluation on the GPU box: C3 (1e6 tracks, 3 states, 46 buckets), a C1-like small dataset and C2;
EXTRACK_TH_DEBUG=1 prints the plan statistics and the apply geometry.  Run under rocprofv3 --kernel-trace --stats for the plan / apply split."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("EXTRACK_TH_DEBUG", "1")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

which = sys.argv[1:] or ["c3", "c1", "c2"]


def run(name, buckets, Ds, Tm, Fs, F, thr=0.2, mx=120, reps=5, F_fw=None):
    ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
    ts = TrackSet(buckets)
    model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, F)
    for _ in range(2):
        tot = ts.loglik_th(model, thr, mx, 2000)
    os.environ.pop("EXTRACK_TH_DEBUG", None)
    t0 = time.perf_counter()
    for _ in range(reps):
        tot = ts.loglik_th(model, thr, mx, 2000)
    wall = (time.perf_counter() - t0) / reps
    ms = ts.ctx.last_kernel_ms()
    fw = []
    for Ff in (F_fw or [F]):
        m2 = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, Ff)
        ts.loglik(m2)
        t0 = time.perf_counter()
        for _ in range(reps):
            ts.loglik(m2)
        fw.append("F=%d %.3f ms" % (Ff, (time.perf_counter() - t0) / reps * 1e3))
    print("%s: %d tracks in %d buckets | th total %.6f wall %.3f ms (events %.3f ms) | fixed window %s" % (
        name, sum(len(b) for b in buckets), len(buckets), tot, wall * 1e3, ms, ", ".join(fw)), flush=True)
    os.environ["EXTRACK_TH_DEBUG"] = "1"
    ts.close()


if "c3" in which:
    Ds, Tm, Fs = [0.0, 0.04, 0.25], [[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]], [.3, .3, .4]
    sizes = synth.bucket_sizes_geometric(1_000_000, list(range(5, 51)), 0.9)
    run("C3", [synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=1000 + L) for L, n in sizes.items() if n > 0], Ds, Tm, Fs, 6, F_fw=[6, 4])
if "c1" in which:
    Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
    sizes = synth.bucket_sizes_geometric(6730, list(range(5, 21)), 0.85)
    run("C1-like", [synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=L) for L, n in sizes.items() if n > 0], Ds, Tm, Fs, 6, reps=20)
if "c2" in which:
    Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
    run("C2", [synth.brownian_tracks(1_000_000, 30, Ds, Tm, Fs, seed=0)], Ds, Tm, Fs, 6)
