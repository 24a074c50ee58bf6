"""Edge cases of the round-2 entry points on the GPU box (not a test: prints what happens)."""
import sys, traceback
import numpy as np
sys.path.insert(0, ".")
from extrack_amd import synth, tracking as T
from extrack_amd.histograms import len_hist
from extrack_amd.refined_localization import position_refinement

def attempt(name, f):
    try:
        r = f()
        print("OK  ", name, "->", r)
    except Exception as e:
        print("FAIL", name, "->", type(e).__name__, str(e)[:200])

Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
def tracks(lens, D=2):
    return {str(L): synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=L)[:, :, :D] for L, n in lens.items()}
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-4, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)

tr = tracks({2: 5, 3: 7, 4: 9, 10: 30})
attempt("len_hist with length-2/3 tracks", lambda: len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=50).shape)
attempt("len_hist 1 track", lambda: len_hist({"8": tr["10"][:1, :8]}, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=50).shape)
attempt("len_hist max_nb_states=1", lambda: float(len_hist(tr, p, 0.02, cell_dims=[1.0], nb_states=2, max_nb_states=1).sum()))
dsr = np.sqrt(2 * np.array([1e-4, 0.25]) * 0.02)
R = lambda trk, **kw: position_refinement(trk, 0.02, dsr, np.array(Fs), np.array(Tm), **kw)
attempt("refinement N=1", lambda: [a.shape for a in R({"10": tr["10"][:1]}, frame_len=6)[0].values()])
attempt("refinement length 2/3/4", lambda: {k: v.shape for k, v in R(tracks({2: 3, 3: 4, 4: 5, 9: 12}), frame_len=6)[0].items()})
attempt("refinement 3-D", lambda: {k: v.shape for k, v in R({"9": np.concatenate([tr["10"][:12, :9], tr["10"][:12, :9, :1]], 2)}, frame_len=6)[0].items()})
attempt("refinement 1-D", lambda: {k: v.shape for k, v in R({"9": tr["10"][:12, :9, :1]}, frame_len=6)[0].items()})
attempt("refinement 5000 tracks, threshold 0.05", lambda: {k: (v.shape, bool(np.isfinite(v).all())) for k, v in R(tracks({20: 5000}), frame_len=6, threshold=0.05)[0].items()})
attempt("refinement NaN position", lambda: float(np.isnan(R({"9": np.where(np.arange(9)[None, :, None] == 4, np.nan, tr["10"][:6, :9])}, frame_len=6)[0]["9"]).sum()))
attempt("fit analytic 1-D tracks", lambda: T.param_fitting({k: v[:, :, :1] for k, v in tracks({6: 80, 12: 60}).items()}, 0.02, params=p, nb_states=2, frame_len=5, cell_dims=[1.0], verbose=0).nfev)
tr3 = {k: np.concatenate([v, v[:, :, :1] * 0.7], 2) for k, v in tracks({6: 80, 12: 60}).items()}
attempt("fit analytic 3-D tracks", lambda: T.param_fitting(tr3, 0.02, params=p, nb_states=2, frame_len=5, cell_dims=[1.0], verbose=0).nfev)
pp = T.generate_params(nb_states=2, LocErr_type=None, estimated_Ds=[1e-4, 0.25], estimated_Fs=[0.6], estimated_transition_rates=0.1, slope_offsets_estimates=[1.0, 0.0]) if "slope_offsets_estimates" in T.generate_params.__code__.co_varnames else None
if pp is not None:
    trl = tracks({6: 80, 12: 60})
    le = {k: np.full(v.shape[:2] + (1,), 0.02) for k, v in trl.items()}
    attempt("fit analytic per-peak LocErr (slope/offset)", lambda: T.param_fitting(trl, 0.02, params=pp, nb_states=2, frame_len=5, cell_dims=[1.0], verbose=0, input_LocErr=le).nfev)
attempt("fit gradient='analytic' with fusion='threshold' (should refuse or fall back)", lambda: T.param_fitting(tracks({6: 80, 12: 60}), 0.02, params=p, nb_states=2, frame_len=5, cell_dims=[1.0], verbose=0, fusion="threshold", gradient="analytic").nfev)
attempt("fit default with fusion='threshold'", lambda: T.param_fitting(tracks({6: 80, 12: 60}), 0.02, params=p, nb_states=2, frame_len=5, cell_dims=[1.0], verbose=0, fusion="threshold").nfev)
attempt("fit nb_substeps=2 analytic", lambda: T.param_fitting(tracks({6: 80, 12: 60}), 0.02, params=p, nb_states=2, nb_substeps=2, frame_len=4, cell_dims=[1.0], verbose=0).nfev)
attempt("fit 3 states Matrix_type 0..4", lambda: [T.param_fitting(tracks({8: 100}), 0.02, nb_states=2, frame_len=4, cell_dims=[1.0], verbose=0, Matrix_type=m).nfev for m in (0, 1, 2, 3, 4)])
