"""Drop-in replacement for ``extrack.refined_localization.position_refinement`` (extrack/refined_localization.py:304-338).

Same name, arguments and result as the reference; the two threshold-fusion passes (get_LC_Km_Ks, :48-204) and the combination of the
predictions from the future and from the past (get_pos_PDF, :207-298) run in HIP kernels through ``extrack_refine_positions`` of the
C ABI.  Built for what the reference's own array reshapes carry through: ONE global localisation error (a float or a 1-element array)
or a dict of per-peak errors ``{len: sigma[n_tracks, len, 1]}``, nb_substeps = 1.  Per-peak errors are used exactly as the reference uses
them - it reverses the error array but not the track in get_LC_Km_Ks (refined_localization.py:64-65 vs :115), so its pass "from the
future" pairs every position with its mirror image's error; that pairing is reproduced (pinned by 50 reference-generated buckets,
tests/golden/refine_pp_cases.*), per-dimension errors and ``[n, len, dims]`` dicts are refused as the reference's reshapes refuse them.
Like the reference, every length bucket is processed as one chunk: its first 30 tracks decide which state sequences are merged."""
import numpy as np

from .engine import TrackSet

__all__ = ["position_refinement", "get_pos_PDF"]


def position_refinement(all_tracks, LocErr, ds, Fs, TrMat, frame_len=7, threshold=0.1, max_nb_states=1000, device=0):
    """all_tracks: {str(len): ndarray[n_tracks, len, dims]}; LocErr: localisation error (std); ds: diffusion lengths sqrt(2 D dt);
    Fs: initial fractions; TrMat: per-step transition probabilities.  Returns ({len: refined positions [n, len, dims]},
    {len: refined stds [n, len]})."""
    per_peak = isinstance(LocErr, dict)
    if not per_peak:
        le = np.atleast_1d(np.asarray(LocErr, dtype=np.float64)).ravel()
        if len(le) != 1:
            raise ValueError("position refinement takes one global localisation error (float) or a dict of per-peak errors "
                             "{len: [n_tracks, len, 1]} (the reference's reshapes assume it, extrack/refined_localization.py:276)")
    print("LocErr_type", "dict" if per_peak else "array")
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
        sig = None
        if per_peak:
            sig = np.asarray(LocErr[l], dtype=np.float64)
            if sig.shape != (Cs.shape[0], Cs.shape[1], 1):
                raise ValueError("per-peak localisation errors must be arrays [n_tracks, len, 1] matching all_tracks[%r]" % l)
        ts = TrackSet([Cs], [sig] if per_peak else None, device=device)
        try:
            model = ts.make_model(None if per_peak else le[None, None], ds, Fs, TrMat, 0.0, [], 1, frame_len)
            all_mus[l], all_sigmas[l] = ts.ctx.refine_positions(model, 0, threshold, max_nb_states)
        finally:
            ts.close()
    return all_mus, all_sigmas


def get_pos_PDF(Cs, LocErr, ds, Fs, TrMat, frame_len=7, threshold=0.2, max_nb_states=1000, device=0):
    """Mirror of extrack/refined_localization.py:207-298 for one array of tracks ``Cs[n_tracks, len, dims]``: the Gaussian mixture that
    describes every position given all the others.  ``LocErr``: the 3-D array position_refinement hands over (``[[[error]]]`` or per-peak
    ``[n_tracks, len, 1]``) or a float.  Returns ``(all_pos_means, all_pos_stds, all_pos_weights)``: lists over the positions of
    ``[n_tracks, n_comp, dims]``, ``[n_tracks, n_comp, 1]``, ``[n_tracks, n_comp]`` (log-weights), components in the reference's order.
    Meant for inspection: every component of every track is copied to the host (position_refinement never materialises them)."""
    Cs = np.asarray(Cs, dtype=np.float64)
    if Cs.ndim != 3 or Cs.shape[1] < 2 or len(Cs) == 0:
        raise ValueError("Cs must be a non-empty array [n_tracks, len >= 2, dims]")
    le = np.asarray(LocErr, dtype=np.float64)
    per_peak = le.ndim == 3 and le.shape[1] == Cs.shape[1] and le.shape[1] > 1
    if per_peak:
        if le.shape != (Cs.shape[0], Cs.shape[1], 1):
            raise ValueError("per-peak localisation errors must be an array [n_tracks, len, 1] matching Cs")
    elif le.size != 1:
        raise ValueError("LocErr must be one global localisation error or per-peak errors [n_tracks, len, 1]")
    ts = TrackSet([Cs], [le] if per_peak else None, device=device)
    try:
        model = ts.make_model(None if per_peak else le.reshape(1, 1, 1), np.asarray(ds, float), np.asarray(Fs, float), np.asarray(TrMat, float), 0.0, [], 1,
                              frame_len)
        return ts.ctx.refine_pos_pdf(model, 0, threshold, max_nb_states)
    finally:
        ts.close()
