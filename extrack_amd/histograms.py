"""Drop-in replacement for the state-duration histogram path of ``extrack.histograms``.

  len_hist        extrack/histograms.py:294-373
  P_segment_len   extrack/histograms.py:26-286

Same names, argument meaning and result as the reference; the enumeration / top-``max_nb_states`` pruning / run-length statistics run in
a hand-written HIP kernel (extrack_amd/csrc/xt_hist.h) through ``extrack_segment_len_hist`` of the C ABI - nothing here computes
probabilities on the CPU.  Differences (documented, corner cases): ``nb_substeps`` must be 1 (with more substeps the reference mixes
substep states into the per-position histories; with isBL it raises); the reference's overflow guards for very long tracks
(chunk-wide shifts, histograms.py:191-192, 244-245) are replaced by a per-track normalisation; ``workers`` is accepted and ignored;
the length buckets are taken in numeric key order (the reference takes the dict's insertion order and only works when that is
ascending).
"""
import numpy as np

from . import engine
from .engine import TrackSet
from .tracking import _resolve_device, extract_params

__all__ = ["len_hist", "P_segment_len"]


def P_segment_len(Cs, LocErr, ds, Fs, TrMat, min_l=3, pBL=0.1, isBL=1, cell_dims=[0.5], nb_substeps=1, max_nb_states=1000, device=0):
    """Mirror of extrack/histograms.py:26-286 for one chunk.  Returns ``(None, None, seg_len_hist)``: the reference's first two return
    values (the per-sequence log-probabilities and state histories of the surviving sequences) never leave the GPU.
    ``seg_len_hist[k - 1, s]`` = expected number of segments of k consecutive positions in state s, summed over the chunk."""
    if nb_substeps != 1:
        raise NotImplementedError("state-duration histograms are built for nb_substeps == 1")
    Cs = np.asarray(Cs, dtype=np.float64)
    if Cs.ndim != 3:
        raise ValueError("Cs must be [n_tracks, len, dims]")
    L = Cs.shape[1]
    if L < 2:
        raise ValueError("minimal track length = 2, here track length = %s" % L)
    LocErr = np.asarray(LocErr, dtype=np.float64)
    if LocErr.ndim != 3 or LocErr.shape[1] not in (1, L):
        raise ValueError("Localization error is not specified correctly, in case of unique localization error specify a float "
                         "number in estimated_vals['LocErr'].")  # histograms.py:60
    per_peak = LocErr.shape[1] == L and L != 1
    if per_peak and LocErr.shape[0] != Cs.shape[0]:
        LocErr = np.broadcast_to(LocErr, (Cs.shape[0],) + LocErr.shape[1:])
    cell_dims = [c for c in np.atleast_1d(np.array(cell_dims, dtype=object)) if c is not None]
    ts = TrackSet([Cs], [LocErr] if per_peak else None, device=device, min_len=max(int(min_l), 1), max_len=(L + 1 if isBL else L))
    try:
        model = ts.make_model(None if per_peak else LocErr, ds, Fs, TrMat, pBL, cell_dims, 1, 2)
        model.c.min_len = int(min_l)
        return None, None, ts.ctx.segment_len_hist(model, 0, max_nb_states)
    finally:
        ts.close()


def len_hist(all_tracks, params, dt, cell_dims=[0.5, None, None], nb_states=2, max_nb_states=500, workers=1, nb_substeps=1, input_LocErr=None,
             device=None, comm=None):
    """State-duration histograms of a length-bucketed dataset (extrack/histograms.py:294-373): ``seg_len_hists[k - 1, s]`` = expected
    number of segments of k consecutive positions in state s over all tracks; shape [longest track length, nb_states].
    The longest bucket is the one whose tracks did not disappear (isBL = 0, histograms.py:335-338); ``min_l`` is the smallest key.
    ``comm`` (extrack_amd.distributed.Comm): every rank processes its row range of every bucket, the histograms are summed over
    the ranks (one all-reduce of the small array)."""
    if nb_substeps != 1:
        raise NotImplementedError("state-duration histograms are built for nb_substeps == 1")
    device = _resolve_device(device, comm)
    keys, tracks, sigmas = engine.sort_buckets(all_tracks, input_LocErr)
    if not tracks:
        raise ValueError("No track could be detected. The loaded tracks seem empty.")
    min_l, max_l = int(keys[0]), int(keys[-1])
    LocErr, ds, Fs, TrMat, pBL = extract_params(params, dt, nb_states, nb_substeps, None)
    S = len(ds)
    cell = [c for c in np.atleast_1d(np.array(cell_dims, dtype=object)) if c is not None]
    if comm is not None:
        from .distributed import shard_range
        sl = [slice(*shard_range(len(b), comm.rank, comm.world)) for b in tracks]
        tracks = [b[s] for b, s in zip(tracks, sl)]
        sigmas = None if sigmas is None else [g[s] for g, s in zip(sigmas, sl)]
        keep = [i for i, b in enumerate(tracks) if len(b)]
        tracks = [tracks[i] for i in keep]
        sigmas = None if sigmas is None else [sigmas[i] for i in keep]
    out = np.zeros((max_l, S))
    if tracks:
        ts = TrackSet(tracks, sigmas, device=device, min_len=min_l, max_len=max_l)
        try:
            if sigmas is not None:
                so = (params["slope_LocErr"].value, params["offset_LocErr"].value) if "slope_LocErr" in params else None
                model = ts.make_model(None, ds, Fs, TrMat, pBL, cell, 1, 2, slope_offset=so)
            else:
                model = ts.make_model(LocErr[0], ds, Fs, TrMat, pBL, cell, 1, 2)
            print("number of chunks:", int(sum(np.ceil(len(b) / 50) for b in tracks)))
            for i, b in enumerate(tracks):
                h = ts.ctx.segment_len_hist(model, i, max_nb_states)
                out[:h.shape[0]] += h
        finally:
            ts.close()
    if comm is not None:
        out = comm.allreduce_vector(out.ravel()).reshape(out.shape)
    print("")
    return out
