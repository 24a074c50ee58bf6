"""Diagnostic: param_fitting(fusion='threshold') on the C1 sim_FOV fixture with the frozen-plan gradient vs finite differences."""
import contextlib, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import tracking as T
info = json.load(open("tests/golden/c1_simfov_10k.json"))
data = np.load("tests/golden/c1_simfov_10k.npz")
tracks = {k: data["tr_" + k] for k in info["keys"]}
for grad in ("analytic", "fd"):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        r = T.param_fitting(tracks, 0.02, nb_states=2, frame_len=6, cell_dims=[1], verbose=1, fusion="threshold", gradient=grad)
    dt = time.perf_counter() - t0
    lines = [l for l in buf.getvalue().split("\n") if l.strip().startswith("-") or l.strip()[:1].isdigit()]
    print(grad, "%.2fs" % dt, "nfev", r.nfev, "ngev", getattr(r, "ngev", 0), "res", r.residual[0], "success", getattr(r, "success", None), "|", getattr(r, "message", ""))
    print({k: round(r.params[k].value, 5) for k in r.params})
    for l in lines[:60]:
        print("   ", l[:160])
