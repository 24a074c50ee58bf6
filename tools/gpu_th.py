"""Timing of the threshold-fusion evaluation (plan + apply kernels) at BASELINE sizes on the GPU box."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402


def run(name, N, L, Ds, Tm, Fs, F, ns, thr, mx, reps=5):
    X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
    ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
    ts = TrackSet([X])
    model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), ns, F)
    tot = ts.loglik_th(model, thr, mx, 2000)
    t0 = time.perf_counter()
    for _ in range(reps):
        tot = ts.loglik_th(model, thr, mx, 2000)
    wall = (time.perf_counter() - t0) / reps
    ms = ts.ctx.last_kernel_ms()
    fw0 = time.perf_counter()
    fw = ts.loglik(model) if 2 ** 0 and len(Ds) ** F <= 8192 else float("nan")
    fw_wall = time.perf_counter() - fw0
    print("%s: N=%d L=%d S=%d F=%d ns=%d thr=%.2f max=%d | th total %.6f wall %.2f ms (events %.2f ms) | fixed-window total %.6f (%.2f ms) | launch %s"
          % (name, N, L, len(Ds), F, ns, thr, mx, tot, wall * 1e3, ms, fw, fw_wall * 1e3, ts.ctx.last_launch_info()), flush=True)
    ts.close()


if __name__ == "__main__":
    run("C2", 1_000_000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], 6, 1, 0.2, 120)
    run("S3", 1_000_000, 30, [0.0, 0.04, 0.25], [[.9, .05, .05], [.05, .9, .05], [.05, .05, .9]], [.3, .3, .4], 6, 1, 0.2, 120)
    run("S4", 500_000, 60, [0.0, 0.02, 0.1, 0.5], [[.85, .05, .05, .05], [.05, .85, .05, .05], [.05, .05, .85, .05], [.05, .05, .05, .85]],
        [.25] * 4, 5, 1, 0.1, 200, reps=2)


def run_multi(name, total, lengths, Ds, Tm, Fs, F, thr, mx, reps=3):
    """Many length buckets (a real dataset has one per track length): the buckets' kernels overlap on a stream pool."""
    sizes = synth.bucket_sizes_geometric(total, lengths, 0.9)
    buckets = [synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=L) for L, n in sizes.items() if n > 0]
    ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
    ts = TrackSet(buckets)
    model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, F)
    for _ in range(2):
        tot = ts.loglik_th(model, thr, mx, 2000)
    t0 = time.perf_counter()
    for _ in range(reps):
        tot = ts.loglik_th(model, thr, mx, 2000)
    wall = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    fw = ts.loglik(model)
    fw_wall = time.perf_counter() - t0
    print("%s: %d tracks in %d buckets (len %d..%d) S=%d F=%d | th total %.6f wall %.2f ms | fixed-window total %.6f (%.2f ms)"
          % (name, total, len(buckets), min(lengths), max(lengths), len(Ds), F, tot, wall * 1e3, fw, fw_wall * 1e3), flush=True)
    ts.close()


if __name__ == "__main__":
    run_multi("C3", 1_000_000, list(range(5, 51)), [0.0, 0.04, 0.25], [[.9, .05, .05], [.05, .9, .05], [.05, .05, .9]], [.3, .3, .4], 4, 0.2, 120)
    run_multi("C1-like", 7_000, list(range(5, 21)), [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], 6, 0.2, 120, reps=10)
