"""A few threshold-fusion evaluations of the C2-size input (for rocprofv3 counter passes)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import synth  # noqa: E402
from extrack_amd.engine import TrackSet  # noqa: E402

N, L = 1_000_000, 30
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
ts = TrackSet([X])
model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, 6)
for _ in range(3):
    print(ts.loglik_th(model, 0.2, 120, 2000))
ts.close()
