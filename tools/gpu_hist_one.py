"""One len_hist call at the reference's default max_nb_states (profiling target)."""
import contextlib, io, sys
sys.path.insert(0, ".")
from extrack_amd import synth, tracking as T
from extrack_amd.histograms import len_hist
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-4, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
tr = {"30": synth.brownian_tracks(100000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1)}
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(2):
        h = len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=500)
print(h.sum())
