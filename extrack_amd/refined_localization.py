"""Drop-in replacement for ``extrack.refined_localization.position_refinement`` (extrack/refined_localization.py:304-338).

Same name, arguments and result as the reference; the two threshold-fusion passes (get_LC_Km_Ks, :48-204) and the combination of the
predictions from the future and from the past (get_pos_PDF, :207-298) run in HIP kernels through ``extrack_refine_positions`` of the
C ABI.  Built for what the reference's own array reshapes support: ONE global localisation error (a float or a 1-element array) and
nb_substeps = 1; per-peak error dicts are refused (the reference pairs them with the wrong positions in its time-reversed pass,
refined_localization.py:69-70).  Like the reference, every length bucket is processed as one chunk: its first 30 tracks decide
which state sequences are merged."""
import numpy as np

from .engine import TrackSet

__all__ = ["position_refinement"]


def position_refinement(all_tracks, LocErr, ds, Fs, TrMat, frame_len=7, threshold=0.1, max_nb_states=1000, device=0):
    """all_tracks: {str(len): ndarray[n_tracks, len, dims]}; LocErr: localisation error (std); ds: diffusion lengths sqrt(2 D dt);
    Fs: initial fractions; TrMat: per-step transition probabilities.  Returns ({len: refined positions [n, len, dims]},
    {len: refined stds [n, len]})."""
    if isinstance(LocErr, dict):
        raise NotImplementedError("position refinement is built for one global localisation error (float), not per-peak error dicts")
    le = np.atleast_1d(np.asarray(LocErr, dtype=np.float64)).ravel()
    if len(le) != 1:
        raise NotImplementedError("position refinement is built for one global localisation error (the reference's reshapes assume it, "
                                  "extrack/refined_localization.py:276)")
    print("LocErr_type", "array")
    ds, Fs, TrMat = np.asarray(ds, float), np.asarray(Fs, float), np.asarray(TrMat, float)
    S = len(ds)
    all_mus, all_sigmas = {}, {}
    for l, Cs in all_tracks.items():
        Cs = np.asarray(Cs, dtype=np.float64)
        if Cs.ndim != 3 or Cs.shape[1] != int(l):
            raise ValueError("all_tracks[%r] must be an array [n_tracks, %s, dims]" % (l, l))
        if len(Cs) == 0:
            all_mus[l], all_sigmas[l] = np.zeros((0, int(l), Cs.shape[2])), np.zeros((0, int(l)))
            continue
        ts = TrackSet([Cs], device=device)
        try:
            model = ts.make_model(le[None, None], ds, Fs, TrMat, 0.0, [], 1, frame_len)
            all_mus[l], all_sigmas[l] = ts.ctx.refine_positions(model, 0, threshold, max_nb_states)
        finally:
            ts.close()
    return all_mus, all_sigmas
