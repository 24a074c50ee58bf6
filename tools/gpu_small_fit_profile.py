"""Where does the time of an experiment-size fit go?  cProfile of param_fitting on the C1 sim_FOV fixture (6 730 tracks), both fusions.
usage: gpu_small_fit_profile.py [window|threshold]"""
import contextlib, cProfile, io, json, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import tracking as T
info = json.load(open("tests/golden/c1_simfov_10k.json"))
data = np.load("tests/golden/c1_simfov_10k.npz")
tracks = {k: data["tr_" + k] for k in info["keys"]}
fusion = sys.argv[1] if len(sys.argv) > 1 else "threshold"
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return T.param_fitting(tracks, 0.02, nb_states=2, frame_len=6, cell_dims=[1], verbose=0, fusion=fusion)
run()  # warm: library load, allocations
t0 = time.perf_counter(); r = run(); dt = time.perf_counter() - t0
print(fusion, "fit %.3f s  nfev %d ngev %d  -> %.1f ms per call" % (dt, r.nfev, getattr(r, "ngev", 0), 1e3 * dt / (r.nfev + getattr(r, "ngev", 0))))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
