"""tools/gpu_th_one.py for A/B runs against an older library build (missing newer entry points are tolerated)."""
import ctypes, sys
_orig = ctypes.CDLL.__getattr__
def _ga(self, name):
    try:
        return _orig(self, name)
    except AttributeError:
        class D: pass
        return D()
ctypes.CDLL.__getattr__ = _ga
import numpy as np
sys.path.insert(0, ".")
from extrack_amd import synth
from extrack_amd.engine import TrackSet
N, L = 1_000_000, 30
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
X = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=0)
ds = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
ts = TrackSet([X])
model = ts.make_model(np.array([[[0.02]]]), ds, np.array(Fs), np.array(Tm), 0.1, (1.0,), 1, 6)
for _ in range(10):
    v = ts.loglik_th(model, 0.2, 120, 2000)
print(v, ts.ctx.last_launch_info())
ts.close()
