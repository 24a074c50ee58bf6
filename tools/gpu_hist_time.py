import contextlib, io, sys, time
sys.path.insert(0, ".")
from extrack_amd import synth, tracking as T
from extrack_amd.histograms import len_hist
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-4, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
tr = {"30": synth.brownian_tracks(100000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1)}
for K in (500, 120):
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            t0 = time.perf_counter(); h = len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=K); dt = time.perf_counter() - t0
    print("K=%d: %.1f ms  sum %.6f" % (K, dt * 1e3, h.sum()), flush=True)
