"""C4-size input (1e7 tracks x 30, 2 states) on ONE GPU: threshold-fusion and fixed-window evaluation times, memory footprint check."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
t0 = time.time()
X = synth.brownian_tracks(N, 30, Ds, Tm, Fs, seed=0)
print("data %.1f s, %.2f GB" % (time.time() - t0, X.nbytes / 1e9), flush=True)
ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
t0 = time.time()
ts = TrackSet([X])
print("upload %.2f s" % (time.time() - t0), flush=True)
model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, 6)
for name, f in (("threshold fusion", lambda: ts.loglik_th(model, 0.2, 120, 2000)), ("fixed window", lambda: ts.loglik(model))):
    f()
    t0 = time.perf_counter()
    for _ in range(3):
        tot = f()
    print("%s: N=%d total %.4f  %.2f ms per evaluation (kernels %.2f ms)" % (name, N, tot, (time.perf_counter() - t0) / 3 * 1e3, ts.ctx.last_kernel_ms()),
          flush=True)
tot_th, ll = ts.loglik_th(model, 0.2, 120, 2000, per_track=True)
print("per-track finite:", bool(np.isfinite(ll).all()), "sum check", abs(ll.sum() - tot_th) / abs(tot_th))
ts.close()
