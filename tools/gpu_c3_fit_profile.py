"""cProfile of the full C3 threshold-fusion fit (1e6 tracks, 3 states): where the wall time goes besides the kernels."""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
Ds = [0.0, 0.04, 0.25]
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
sizes = synth.bucket_sizes_geometric(int(1e6), list(range(5, 51)), 0.9)
tracks = {str(L): synth.brownian_tracks(n, L, Ds, Tm, [0.3, 0.3, 0.4], seed=L) for L, n in sizes.items() if n > 0}
p0 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4], estimated_Fs=[0.33, 0.33], estimated_transition_rates=0.1)
fusion = sys.argv[1] if len(sys.argv) > 1 else "threshold"
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return T.param_fitting(tracks, 0.02, params=p0, nb_states=3, nb_substeps=1, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], threshold=0.2, max_nb_states=120, fusion=fusion)
run()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); r = run(); pr.disable(); dt = time.perf_counter() - t0
print(fusion, "fit %.2f s nfev %d ngev %d" % (dt, r.nfev, getattr(r, "ngev", 0)))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
