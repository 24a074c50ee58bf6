"""Throughput of the state-duration histograms (len_hist) and of the position refinement on the GPU box."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, ".")
from extrack_amd import synth, tracking as T
from extrack_amd.histograms import len_hist
from extrack_amd.refined_localization import position_refinement

Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-4, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
ds = np.sqrt(2 * np.array([1e-4, 0.25]) * 0.02)
for N, L in ((100_000, 30), (20_000, 60)):
    tr = {str(L): synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=1)}
    for K in (120, 500):
        with contextlib.redirect_stdout(io.StringIO()):
            len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=K)
            t0 = time.perf_counter()
            h = len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=K)
            dt = time.perf_counter() - t0
        print("len_hist   %7d tracks x %d, max_nb_states %3d: %.3f s = %.3g tracks/s (hist sum %.1f)" % (N, L, K, dt, N / dt, h.sum()), flush=True)
    with contextlib.redirect_stdout(io.StringIO()):
        position_refinement(tr, 0.02, ds, np.array(Fs), np.array(Tm), frame_len=6, threshold=0.1, max_nb_states=200)
        t0 = time.perf_counter()
        mu, sg = position_refinement(tr, 0.02, ds, np.array(Fs), np.array(Tm), frame_len=6, threshold=0.1, max_nb_states=200)
        dt = time.perf_counter() - t0
    print("refinement %7d tracks x %d: %.3f s = %.3g tracks/s" % (N, L, dt, N / dt), flush=True)
