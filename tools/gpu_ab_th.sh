cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in var_9f02.so libextrack_hip.so; do
    EXTRACK_HIP_LIB=$PWD/extrack_amd/$lib python - <<'PY'
import os, sys, time, ctypes
_orig = ctypes.CDLL.__getattr__
def _ga(self, name):
    try:
        return _orig(self, name)
    except AttributeError:
        class D: pass
        return D()
ctypes.CDLL.__getattr__ = _ga  # an older library lacks the newest entry points: A/B of what both have
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
for _ in range(3): v = ts.loglik_th(model, 0.2, 120, 2000)
ks = []
for _ in range(10):
    v = ts.loglik_th(model, 0.2, 120, 2000); ks.append(ts.ctx.last_kernel_ms())
print(os.path.basename(os.environ["EXTRACK_HIP_LIB"]), "th kernels ms %.3f" % np.mean(ks), v)
ts.close()
PY
  done
done
