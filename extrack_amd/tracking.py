"""Drop-in replacement for the track-likelihood path of ``extrack.tracking``.

Same public names, argument meaning and error behaviour as the reference for THIS path
(reference file:line relative to the reference root):

  param_fitting     extrack/tracking.py:1299-1386  (fixed-window twin: extrack/tracking_0.py:918-1022)
  predict_Bs        extrack/tracking.py:792-906    (twin: extrack/tracking_0.py:463-563)
  cum_Proba_Cs      extrack/tracking.py:991-1088   (twin: extrack/tracking_0.py:637-715)
  Proba_Cs          extrack/tracking_0.py:440-458
  P_Cs_inter_bound_stats  extrack/tracking.py:109-318
  extract_params    extrack/tracking.py:913-986
  generate_params   extrack/tracking.py:1214-1290
  get_params        extrack/tracking.py:1090-1212

The recursion itself runs in hand-written HIP kernels for gfx950 through the C ABI in
include/extrack_hip.h; nothing here computes likelihoods on the CPU.  Two kernels exist, selected by the extra
keyword ``fusion``:
  ``fusion="window"`` (default)  the FIXED-WINDOW kernel ``P_Cs_inter_bound_stats`` (tracking.py:109-318, what
                                 BASELINE.json names); ``threshold`` / ``max_nb_states`` are accepted and ignored.
  ``fusion="threshold"``         the THRESHOLD-FUSION kernel ``P_Cs_inter_bound_stats_th`` (tracking.py:427-743) that
                                 ``extrack.tracking.param_fitting`` calls in v1.6.3, chunked by 2000 tracks like
                                 ``cum_Proba_Cs`` (tracking.py:1043); ``predict_Bs`` then works in chunks of ``nb_max``
                                 tracks (tracking.py:856-868).
``workers`` is accepted and ignored: the tracks are sharded over GPUs instead (extrack_amd.distributed).
"""
import os
import sys
import time

import numpy as np
from scipy import linalg

from . import _lib, engine
from .engine import TrackSet
from .lmfit_compat import Parameters, is_parameters, minimize

__all__ = ["param_fitting", "predict_Bs", "cum_Proba_Cs", "cum_Proba_Cs_grad", "Proba_Cs", "P_Cs_inter_bound_stats", "extract_params",
           "generate_params", "get_params", "TrackSet"]


# ------------------------------------------------------------------------------------------------------------
# parameter plumbing
# ------------------------------------------------------------------------------------------------------------
def _extract_arrays(params, dt, nb_substeps, Matrix_type=1):
    """The arithmetic of extract_params on any mapping ``{name: object with .value}`` - real or COMPLEX values (the gradient path
    differentiates it by the complex-step method, extrack_amd/gradient.py).  Returns (global LocErr [k], Ds [S], Fs [S],
    TrMat [S, S], pBL, (slope, offset) or None)."""
    names = np.sort(list(params.keys()))
    le = np.array([params[n].value for n in names if n.startswith("LocErr")])
    so = (params["slope_LocErr"].value, params["offset_LocErr"].value) if "slope_LocErr" in params else None
    Ds = np.array([params[n].value for n in names if n.startswith("D") and len(n) < 3])
    Fs = np.array([params[n].value for n in names if n.startswith("F")])
    S = len(Ds)
    cplx = any(isinstance(params[n].value, complex) for n in params)
    TrMat = np.zeros((S, S), dtype=complex if cplx else float)
    pBL = None
    for n in params:
        if n == "pBL":
            pBL = params[n].value
        elif n.startswith("p"):
            TrMat[int(n[1]), int(n[2])] = params[n].value
    TrMat = TrMat / nb_substeps
    diag = np.arange(S)
    if Matrix_type == 0:
        TrMat[diag, diag] = 1 - np.sum(TrMat, 1)
    if Matrix_type == 1:
        TrMat = 1 - np.exp(-TrMat)
        TrMat[diag, diag] = 1 - np.sum(TrMat, 1)
    elif Matrix_type == 2:
        TrMat[diag, diag] = -np.sum(TrMat, 1)
        TrMat = linalg.expm(TrMat)
    elif Matrix_type in (3, 4):
        TrMat[diag, diag] = 0
        G = np.copy(TrMat)
        TrMat[diag, diag] = 1 - np.sum(TrMat, 1)
        G[diag, diag] = -np.sum(G, 1)
        TrMatG = linalg.expm(G)
        TrMat = np.mean([TrMat, TrMatG], axis=0) if Matrix_type == 3 else (TrMat * TrMatG) ** 0.5
    return le, Ds, Fs, TrMat, pBL, so


def extract_params(params, dt, nb_states, nb_substeps, input_LocErr=None, Matrix_type=1):
    """Parameters -> (LocErr, ds, Fs, TrMat, pBL); mirrors extrack/tracking.py:913-986.

    ``nb_states`` is accepted and unused, exactly like the reference.  ``LocErr`` is a list: one
    (1,1,k) array for a global error, or one per-peak array per bucket when ``input_LocErr`` is given
    (affinely rescaled and clipped at 1e-6 if ``slope_LocErr``/``offset_LocErr`` are parameters)."""
    if isinstance(dt, dict):
        raise TypeError("dt as a dict is sorted into a list (bucket order) by param_fitting / predict_Bs before extract_params "
                        "(extrack/tracking.py:1349-1368): pass that list")
    le, Ds, Fs, TrMat, pBL, so = _extract_arrays(params, dt, nb_substeps, Matrix_type)
    LocErr = [le[None, None]]
    if input_LocErr is not None:
        if so is not None:
            LocErr = [np.clip(x * so[0] + so[1], 0.000001, np.inf) for x in input_LocErr]
        else:
            LocErr = input_LocErr
    if isinstance(dt, list):  # per-track time steps: one array [n_tracks, len, S] per bucket (tracking.py:979-982)
        ds = [np.sqrt(2 * Ds[None, None] * np.asarray(t, float)[:, :, None]) for t in dt]
    else:
        ds = np.sqrt(2 * Ds * dt)
    return LocErr, ds, Fs, TrMat, pBL


def generate_params(nb_states=3, LocErr_type=1, nb_dims=3, LocErr_bounds=[0.005, 0.1], D_max=10, Fractions_bounds=[0.001, 0.99],
                    estimated_LocErr=None, estimated_Ds=None, estimated_Fs=None, estimated_transition_rates=0.1,
                    slope_offsets_estimates=None):
    """Initial Parameters for an n-state model; same names/values/bounds as extrack/tracking.py:1214-1290."""
    rows = []
    for s in range(nb_states):
        v = 0.5 * s ** 2 * D_max / (nb_states - 1) ** 2 if estimated_Ds is None else estimated_Ds[s]
        rows.append(dict(name="D%d" % s, value=v, min=0, max=D_max, vary=True))
    geo = (LocErr_bounds[0] * LocErr_bounds[1]) ** 0.5
    le = (lambda i: geo) if estimated_LocErr is None else (lambda i: estimated_LocErr[i])
    lo, hi = LocErr_bounds
    if LocErr_type == 1:
        rows.append(dict(name="LocErr", value=le(0), min=lo, max=hi, vary=True))
    elif LocErr_type == 2:
        for d in range(nb_dims):
            rows.append(dict(name="LocErr%d" % d, value=le(d), min=lo, max=hi, vary=True))
    elif LocErr_type == 3:
        rows.append(dict(name="LocErr0", value=le(0), min=lo, max=hi, vary=True))
        rows.append(dict(name="LocErr1", expr="LocErr0"))
        rows.append(dict(name="LocErr2", value=le(-1), min=lo, max=hi, vary=True))
    if LocErr_type == 4:
        rows.append(dict(name="slope_LocErr", value=slope_offsets_estimates[0], min=-1, max=20, vary=True))
        rows.append(dict(name="offset_LocErr", value=slope_offsets_estimates[1], min=-1, max=1, vary=True))
    F_expr = "1"
    for s in range(nb_states - 1):
        v = 1 / nb_states if estimated_Fs is None else estimated_Fs[s]
        rows.append(dict(name="F%d" % s, value=v, min=Fractions_bounds[0], max=Fractions_bounds[1], vary=True))
        F_expr += " - F%d" % s
    rows.append(dict(name="F%d" % (nb_states - 1), expr=F_expr))
    if not isinstance(estimated_transition_rates, (np.ndarray, list)):
        estimated_transition_rates = [estimated_transition_rates] * (nb_states * (nb_states - 1))
    idx = 0
    for i in range(nb_states):
        for j in range(nb_states):
            if i != j:
                rows.append(dict(name="p%d%d" % (i, j), value=estimated_transition_rates[idx], min=0.0001, max=1, vary=True))
                idx += 1
    rows.append(dict(name="pBL", value=0.1, min=0.0001, max=1, vary=True))
    params = Parameters()
    for r in rows:
        params.add(**r)
    return params


_GP_VARY = {'LocErr': True, 'D0': True, 'D1': True, 'F0': True, 'p01': True, 'p10': True, 'pBL': True}
_GP_EST = {'LocErr': 0.025, 'D0': 1e-20, 'D1': 0.05, 'F0': 0.45, 'p01': 0.05, 'p10': 0.05, 'pBL': 0.1}
_GP_MIN = {'LocErr': 0.007, 'D0': 1e-12, 'D1': 0.00001, 'F0': 0.001, 'p01': 0.01, 'p10': 0.01, 'pBL': 0.01}
_GP_MAX = {'LocErr': 0.6, 'D0': 1, 'D1': 10, 'F0': 0.999, 'p01': 1., 'p10': 1., 'pBL': 0.99}


def get_params(nb_states=2, steady_state=False, vary_params=_GP_VARY, estimated_vals=_GP_EST, min_values=_GP_MIN, max_values=_GP_MAX):
    """Parameters from explicit dicts; same construction as extrack/tracking.py:1090-1212 (the generic branch is
    the only live one there: D_k are chained through ``D{k}_minus_D{k-1}`` increments, the last fraction is
    ``1-F0-...``).  ``nb_states``/``steady_state`` are accepted and unused, like the reference."""
    rows = []
    if "slope_LocErr" in estimated_vals:
        for n in ("slope_LocErr", "offset_LocErr"):
            rows.append(dict(name=n, value=estimated_vals[n], min=min_values[n], max=max_values[n], vary=vary_params[n]))
    if "LocErr" in estimated_vals:
        LocErr = estimated_vals["LocErr"]
        if type(LocErr) == float:
            rows.append(dict(name="LocErr", value=LocErr, min=min_values["LocErr"], max=max_values["LocErr"], vary=vary_params["LocErr"]))
        elif isinstance(LocErr, (np.ndarray, list)):
            for s in range(len(LocErr)):
                rows.append(dict(name="LocErr%d" % s, value=LocErr[s], min=min_values["LocErr"][s], max=max_values["LocErr"][s],
                                 vary=vary_params["LocErr"][s]))
    Ds = [k for k in vary_params if k.startswith("D")]
    Fs = [k for k in vary_params if k.startswith("F")]
    rows.append(dict(name="D0", value=estimated_vals["D0"], min=min_values["D0"], max=0.3, brute_step=0.04, vary=vary_params["D0"]))
    last_D, sum_Ds, expr = "D0", estimated_vals["D0"], "D0"
    for D in Ds[1:]:
        rows.append(dict(name=D + "_minus_" + last_D, value=estimated_vals[D] - sum_Ds, min=0, max=max_values[D], vary=vary_params[D]))
        expr = expr + "+" + D + "_minus_" + last_D
        rows.append(dict(name=D, expr=expr))
        last_D = D
        sum_Ds += estimated_vals[D]
    rows.append(dict(name="F0", value=estimated_vals["F0"], min=min_values["F0"], max=max_values["F0"], brute_step=0.04,
                     vary=vary_params["F0"]))
    expr = "1-F0"
    for F in Fs[1:len(Ds) - 1]:
        rows.append(dict(name=F, value=estimated_vals[F], min=0.001, max=0.99, vary=vary_params[F]))
        expr = expr + "-" + F
    rows.append(dict(name="F%d" % (len(Ds) - 1), expr=expr))
    for p in vary_params:
        if p.startswith("p"):
            rows.append(dict(name=p, value=estimated_vals[p], min=min_values[p], max=max_values[p], vary=vary_params[p]))
    params = Parameters()
    for r in rows:
        params.add(**r)
    return params


def _sorted_dt(dt, all_tracks):
    """dt given as {len: array[n_tracks, len]} -> list in the order of the non-empty buckets (extrack/tracking.py:1349-1368); a scalar
    dt is returned unchanged."""
    if not isinstance(dt, dict):
        return dt
    keys = np.sort(np.array(list(all_tracks.keys())).astype(int)).astype(str)
    return [np.asarray(dt[k], dtype=np.float64) for k in keys if len(all_tracks[k]) > 0]


def _check_fusion(fusion):
    if fusion is None:
        fusion = default_fusion()
    if fusion not in ("window", "threshold"):
        raise ValueError("fusion must be 'window' or 'threshold'")
    return fusion == "threshold"


# ------------------------------------------------------------------------------------------------------------
# kernel-level mirrors (one-off uploads; used by tests and for drop-in calls on small chunks)
# ------------------------------------------------------------------------------------------------------------
def _one_bucket(Cs, LocErr, isBL, min_len, device):
    Cs = np.asarray(Cs, dtype=np.float64)
    if Cs.ndim != 3:
        raise ValueError("Cs must be [n_tracks, len, dims]")
    L = Cs.shape[1]
    if L < 2:
        raise ValueError("minimal track length = 2, here track length = %s" % L)  # tracking.py:149-150
    LocErr = np.asarray(LocErr, dtype=np.float64)
    if LocErr.ndim != 3 or LocErr.shape[1] not in (1, L):
        raise ValueError("Localization error is not specified correctly, in case of unique localization error specify a float "
                         "number in estimated_vals['LocErr'].")  # tracking.py:143
    per_peak = LocErr.shape[1] == L and not (L == 1)
    if per_peak and LocErr.shape[0] != Cs.shape[0]:
        LocErr = np.broadcast_to(LocErr, (Cs.shape[0],) + LocErr.shape[1:])
    ts = TrackSet([Cs], [LocErr] if per_peak else None, device=device, min_len=max(int(min_len), 2),
                  max_len=(L + 1 if isBL else L))
    return ts, (None if per_peak else LocErr)


def Proba_Cs(Cs, LocErr, ds, Fs, TrMat, pBL, isBL, cell_dims, nb_substeps, frame_len, min_len, threshold=None, max_nb_states=None,
             device=0, fusion="window"):
    """Per-track log-likelihood LP_C[N] of one chunk, computed on the GPU: extrack/tracking_0.py:440-458 (fixed window) or,
    with ``fusion="threshold"``, extrack/tracking.py:769-787 (the whole of ``Cs`` is ONE chunk: its first 30 tracks decide
    the merges, tracking.py:678-679)."""
    ts, le = _one_bucket(Cs, LocErr, isBL, min_len, device)
    try:
        model = ts.make_model(le, ds, Fs, TrMat, pBL, cell_dims, nb_substeps, frame_len)
        if _check_fusion(fusion):
            return ts.loglik_th(model, 0.2 if threshold is None else threshold, 120 if max_nb_states is None else max_nb_states,
                                chunk=max(len(Cs), 1), per_track=True)[1]
        return ts.loglik(model, per_track=True)[1]
    finally:
        ts.close()


def get_all_Bs(nb_Cs, nb_states):
    """Matrix [nb_states**nb_Cs, nb_Cs] of all state sequences: digit k of row i is (i // nb_states**k) % nb_states (k = 0: the newest
    state) - the layout of ``cur_Bs`` and of the columns of ``LP`` in the reference (extrack/tracking.py:746-757)."""
    i = np.arange(int(nb_states) ** int(nb_Cs))
    return np.stack([(i // nb_states ** k) % nb_states for k in range(int(nb_Cs))], axis=1).astype(int)


def P_Cs_inter_bound_stats(Cs, LocErr, ds, Fs, TrMat, pBL=0.1, isBL=1, cell_dims=[0.5], nb_substeps=1, frame_len=4, do_preds=0,
                           min_len=3, device=0, return_matrix=False):
    """Mirror of extrack/tracking.py:109-318.  The likelihood kernels reduce the per-sequence matrix ``LP[N, nB]`` in place, so by
    default the first return value is ``LP_C[:, None]`` (its logsumexp over axis 1 is what Proba_Cs computes from the reference's
    matrix) and ``cur_Bs`` is None.  ``return_matrix=True`` gives the reference's contract, ``(LP[N, nB], cur_Bs[1, nB, n], preds)``
    with the reference's column order (``get_all_Bs``) - for small inputs: the matrix goes through host memory.
    ``preds`` is ``[]`` when ``do_preds`` is 0."""
    ts, le = _one_bucket(Cs, LocErr, isBL, min_len, device)
    try:
        model = ts.make_model(le, ds, Fs, TrMat, pBL, cell_dims, nb_substeps, frame_len)
        preds = []
        if do_preds:
            if nb_substeps != 1:
                raise ValueError("state predictions require nb_substeps == 1")
            preds = ts.predict(model)[0]
        if return_matrix:
            LP = ts.ctx.sequence_matrix(model, 0)
            S = len(np.asarray(ds))
            n = int(round(np.log(LP.shape[1]) / np.log(S)))
            return LP, get_all_Bs(n, S)[None], preds
        lpc = ts.loglik(model, per_track=True)[1]
        return lpc[:, None], None, preds
    finally:
        ts.close()


def P_Cs_inter_bound_stats_th(Cs, LocErr, ds, Fs, TrMat, pBL=0.1, isBL=1, cell_dims=[0.5], nb_substeps=1, frame_len=6, do_preds=0, min_len=3,
                              threshold=0.2, max_nb_states=120, device=0, return_matrix=False):
    """Mirror of extrack/tracking.py:427-650 (the kernel extrack.tracking runs in v1.6.3): the whole of ``Cs`` is ONE chunk, its first 30
    tracks decide which state sequences are merged (:678-679).  Returns ``(LP, cur_Bs_cat, preds)``:
      * ``LP``: by default ``LP_C[:, None]`` - the likelihood kernels reduce the per-sequence matrix in place, and its log-sum over axis 1 is
        all the reference's callers take from it (Proba_Cs, :778-787).  ``return_matrix=True`` gives the reference's ``LP[N, nB]`` itself,
        column by column (the sequences alive after the last merge x the new states of the last step, x the states of the leaving step for
        isBL tracks, :611-633) - for small inputs: the matrix goes through host memory (extrack_sequence_matrix_th).
      * ``cur_Bs_cat`` is None: the reference's state-history array is internal to its grouping rule and read by no caller (:778, :790).
      * ``preds``: the posteriors [N, len, S] when ``do_preds`` (what predict_Bs takes, :790), else ``[]``."""
    ts, le = _one_bucket(Cs, LocErr, isBL, min_len, device)
    try:
        ds, TrMat = np.asarray(ds, float), np.asarray(TrMat, float)
        S, ns = len(ds), int(nb_substeps)
        model = ts.make_model(le, ds, Fs, TrMat, pBL, cell_dims, ns, frame_len)
        N = len(Cs)
        preds = []
        if do_preds:
            if ns != 1:
                raise ValueError("state predictions require nb_substeps == 1")
            preds = ts.predict_th(model, threshold, max_nb_states, nb_max=max(N, 1))[0]
        if not return_matrix:
            return ts.loglik_th(model, threshold, max_nb_states, chunk=max(N, 1), per_track=True)[1][:, None], None, preds
        LP = ts.ctx.sequence_matrix_th(model, 0, threshold, max_nb_states)
        if isBL:
            # the leaving / bleaching step (:611-630): every sequence is expanded once more by the S^ns states of that step, new index =
            # old * S^ns + r2; its factor depends on the model only: the ns transitions from the sequence's newest state through the digits of r2
            # (digit 0 = newest) and the chance to leave the field of view or bleach from r2's newest state - p_stay indexed by the raw
            # state (:624)
            G = S ** ns
            ps = engine.p_stay_table(ds, S, ns, cell_dims)
            r2 = np.arange(G)
            dig = np.stack([(r2 // S ** c) % S for c in range(ns)], 1)  # [G, ns], column 0 = newest
            logT = np.log(TrMat)
            LL = np.zeros((S, G))
            for prev in range(S):
                chain = np.concatenate([dig, np.full((G, 1), prev)], 1)  # newest ... oldest = the sequence's newest state
                for c in range(ns):
                    LL[prev] += logT[chain[:, c + 1], chain[:, c]]
                e = ps[dig[:, 0]]
                LL[prev] += np.log(pBL + (1 - e) - pBL * (1 - e))
            newest = np.arange(LP.shape[1]) % S  # column (g, r): digit 0 of r is the sequence's newest state
            LP = (LP[:, :, None] + LL[newest][None]).reshape(N, -1)
        return LP, None, preds
    finally:
        ts.close()


# ------------------------------------------------------------------------------------------------------------
# objective
# ------------------------------------------------------------------------------------------------------------
def default_fusion():
    """Kernel used when a call does not pass ``fusion=``: "window" (the fixed-window kernel BASELINE.json names) unless the
    environment variable ``EXTRACK_FUSION=threshold`` selects the kernel extrack.tracking runs in v1.6.3 for the whole process."""
    return os.environ.get("EXTRACK_FUSION", "window").strip().lower() or "window"


def _resolve_device(device, comm):
    """GPU index: explicit, else the communicator's device (LOCAL_RANK under torchrun), else 0."""
    if device is not None:
        return int(device)
    return comm.local_device() if comm is not None else 0


def _as_trackset(all_tracks, input_LocErr, device=None, comm=None, dt=None):
    """(TrackSet, owned).  A ``TrackSet`` is used as is.  A list of bucket arrays (what the reference's objective receives at
    every call) is uploaded for THIS call only and released afterwards (``owned``): device copies are never cached behind the
    caller's back, so edited or re-allocated arrays can not be confused with earlier ones.  Keep a ``TrackSet`` (or use
    ``param_fitting``) to pay the upload once.  With ``comm`` the list is this rank's shard and the dataset-global
    min / max length are agreed over the ranks (collective)."""
    if isinstance(all_tracks, TrackSet):
        return all_tracks, False
    dev = _resolve_device(device, comm)
    dts = dt if isinstance(dt, list) else None
    if comm is None:
        return TrackSet(list(all_tracks), input_LocErr, device=dev, dts=dts), True
    if dts is not None:
        raise ValueError("per-track time steps with a communicator need chunk-aligned shards: pass the TrackSet of "
                         "Comm.shard_trackset(..., chunk=max_number_of_tracks_per_matrix, dts=...)")
    lo, hi = comm.global_min_max_len([np.shape(b)[1] for b in all_tracks if len(b)])
    return TrackSet(list(all_tracks), input_LocErr, device=dev, min_len=lo, max_len=hi, allow_empty=True), True


def _objective_model(params, ts, dt, cell_dims, input_LocErr, nb_states, nb_substeps, frame_len, Matrix_type, dt_chunk=None):
    """Model handle of one objective evaluation, or None for invalid parameters (tracking.py:1017).  With per-track time steps
    (``ts.has_dt``; ``dt`` is then ignored, the arrays live with the TrackSet) the diffusion lengths are those of a unit time step
    and the validity check looks at the diffusion coefficients themselves (the reference compares medians of ds, tracking.py:1011-1017:
    the same order)."""
    le, Ds, Fs, TrMat, pBL, so = _extract_arrays(params, dt, nb_substeps, Matrix_type)
    ds = np.sqrt(2 * Ds) if ts.has_dt else np.sqrt(2 * Ds * dt)
    valid = bool(np.all(TrMat > 0) and np.all(Fs > 0) and np.all(ds[1:] - ds[:-1] >= 0))  # tracking.py:1017
    if not valid:
        return None
    if ts.has_sigma:  # per-peak errors win over any LocErr parameter (tracking.py:926-932)
        return ts.make_model(None, ds, Fs, TrMat, pBL, cell_dims, nb_substeps, frame_len, slope_offset=so, dt_chunk=dt_chunk)
    return ts.make_model(le[None, None], ds, Fs, TrMat, pBL, cell_dims, nb_substeps, frame_len, dt_chunk=dt_chunk)


def cum_Proba_Cs(params, all_tracks, dt, cell_dims, input_LocErr, nb_states, nb_substeps, frame_len, verbose=1, workers=1,
                 Matrix_type=1, threshold=0.2, max_nb_states=120, max_number_of_tracks_per_matrix=2000, comm=None, fusion=None,
                 device=None):
    """-sum of per-track log-likelihoods, or +inf for invalid parameters / NaN (extrack/tracking.py:991-1088).

    ``all_tracks``: a ``TrackSet`` (device-resident, reused between calls) or the list of bucket arrays sorted short->long that
    the reference passes (uploaded for this call only).  ``comm``: optional extrack_amd.distributed.Comm; ``all_tracks`` is
    then this rank's shard - preferably the ``TrackSet`` made by ``comm.shard_trackset`` - and the scalar is all-reduced over
    the ranks.  ``fusion="threshold"``: the v1.6.3 kernel with ``threshold``, ``max_nb_states`` and chunks of
    ``max_number_of_tracks_per_matrix`` tracks (the chunking is part of the result: with ``comm`` the shards must be whole chunks,
    ``Comm.shard_trackset(..., chunk=...)``).  ``fusion=None``: the process default (``EXTRACK_FUSION``, else "window")."""
    th = _check_fusion(fusion)
    if th and comm is not None and not isinstance(all_tracks, TrackSet):
        raise ValueError("fusion='threshold' with comm needs chunk-aligned shards: pass the TrackSet of comm.shard_trackset(..., chunk=...)")
    ts, owned = _as_trackset(all_tracks, input_LocErr, device, comm, dt)
    try:
        if ts.has_dt and not th:
            raise NotImplementedError("per-track time steps (dt as a dict / list of arrays) exist in the threshold-fusion kernel only "
                                      "(extrack/tracking.py:494-499); the fixed-window kernel of extrack/tracking_0.py has no such input: "
                                      "use fusion='threshold'")
        model = _objective_model(params, ts, dt, cell_dims, input_LocErr, nb_states, nb_substeps, frame_len, Matrix_type,
                                 dt_chunk=max_number_of_tracks_per_matrix)
        if model is not None:
            if th and comm is not None:  # shards are whole chunks, so the chunking is that of the unsharded dataset
                Cum_P = comm.allreduce_loglik_th(ts, model, threshold, max_nb_states, max_number_of_tracks_per_matrix)
            elif th:
                Cum_P = ts.loglik_th(model, threshold, max_nb_states, max_number_of_tracks_per_matrix)
            else:
                Cum_P = ts.loglik(model) if comm is None else comm.allreduce_loglik(ts, model)
    finally:
        if owned:
            ts.close()
    if model is not None:
        if verbose == 1:
            q = [p + " = " + str(np.round(params[p].value, 6)) for p in params]
            print(Cum_P, q)
        else:
            print(".", end="")
        out = -Cum_P
    else:
        out = np.inf
        print("x", end="")
        if verbose == 1:
            print([p + " = " + str(np.round(params[p].value, 4)) for p in params])
    if np.isnan(out):
        out = np.inf
        print("input parameters give nans, you may want to pick more suitable parameter initial values")
    return out


def cum_Proba_Cs_grad(params, names, all_tracks, dt, cell_dims, input_LocErr, nb_states, nb_substeps, frame_len, verbose=1, workers=1,
                      Matrix_type=1, threshold=0.2, max_nb_states=120, max_number_of_tracks_per_matrix=2000, comm=None, fusion=None,
                      device=None):
    """(objective, gradient): ``cum_Proba_Cs`` and its exact derivative with respect to the VALUES of the free parameters ``names``,
    from ONE pass of the gradient kernels instead of the nvar + 1 evaluations the reference's optimiser spends on finite differences
    (extrack/tracking.py:1371).  ``fusion="window"``: extrack_loglik_grad; ``fusion="threshold"`` (what extrack.tracking.param_fitting
    minimises in v1.6.3): extrack_loglik_th_grad - the value is that of ``cum_Proba_Cs(..., fusion="threshold")`` and the gradient its
    derivative with the merge groups of THIS evaluation held fixed (the plan is re-decided at every call, exactly as the objective does).
    Same argument list as ``cum_Proba_Cs`` after ``names``; same prints; (+inf, zeros) for invalid parameters or NaN."""
    from . import gradient
    th = _check_fusion(fusion)
    if th and comm is not None and not isinstance(all_tracks, TrackSet):
        raise ValueError("fusion='threshold' with comm needs chunk-aligned shards: pass the TrackSet of comm.shard_trackset(..., chunk=...)")
    ts, owned = _as_trackset(all_tracks, input_LocErr, device, comm)
    try:
        out, g = gradient.objective_and_gradient(params, ts, dt, cell_dims, nb_states, nb_substeps, frame_len, Matrix_type, comm, names,
                                                 threshold_fusion=(threshold, max_nb_states, max_number_of_tracks_per_matrix) if th else None)
    finally:
        if owned:
            ts.close()
    if np.isfinite(out):
        if verbose == 1:
            q = [p + " = " + str(np.round(params[p].value, 6)) for p in params]
            print(-out, q)
        else:
            print(".", end="")
    elif np.isnan(out):
        out, g = np.inf, np.zeros(len(g))
        print("input parameters give nans, you may want to pick more suitable parameter initial values")
    else:
        print("x", end="")
        if verbose == 1:
            print([p + " = " + str(np.round(params[p].value, 4)) for p in params])
    return out, g


def _pick_gradient(params, fargs, explicit, comm=None, info=None):
    """Should ``param_fitting`` hand the optimiser the analytic gradient (one gradient-kernel pass per iteration) rather than let it
    difference the objective (nvar + 1 evaluations per iteration, what the reference does, extrack/tracking.py:1371)?
    Yes when (a) every constraint expression is complex-differentiable and (b) - unless the caller asked for it explicitly - a timing
    probe on this dataset says a gradient call costs less than the nvar + 1 objective calls it replaces: on large datasets the gradient
    kernels beat finite differences (two states: tangents in registers, xt_reg2.h; 3 / 4 states: reverse mode, xt_rev.h; threshold
    fusion: reverse mode at the frozen plan, xt_thgrad.h - a few objective calls' worth whatever nvar is), on datasets of a few thousand
    tracks the comparison is decided by host overheads.  With ``comm`` the decision is taken on the slowest rank's timings.
    ``info`` (dict) receives ``gradient_path`` ("analytic" | "fd") and ``gradient_why``: the decision is part of the fit's record.  Only a
    model the gradient kernels do not serve (ExtrackError E_UNSUPPORTED / NotImplementedError) falls back silently; any other failure of
    the probe is an error of the gradient path and is raised."""
    import time
    from . import gradient
    info = {} if info is None else info

    def decide(use, why):
        info["gradient_path"], info["gradient_why"] = ("analytic" if use else "fd"), why
        return use
    names = gradient.free_names(params)
    why = gradient.analytic_support(params, names)
    if why is not None:
        if explicit:
            raise ValueError("gradient='analytic': a parameter expression is not differentiable (%s)" % why)
        return decide(False, "a parameter expression is not complex-differentiable (%s)" % why)
    if explicit:
        return decide(True, "requested (gradient='analytic')")
    import contextlib, io
    a = list(fargs)
    a[7] = 0  # verbose
    sink = io.StringIO()
    unsupported = None
    t_ll = t_g = 0.0
    with contextlib.redirect_stdout(sink):
        cum_Proba_Cs(params, *a)  # warm-up: tables, workspaces, clocks
        t0 = time.perf_counter()
        cum_Proba_Cs(params, *a)
        t_ll = time.perf_counter() - t0
        try:
            t0 = time.perf_counter()
            cum_Proba_Cs_grad(params, names, *a)  # first call: includes one-time allocations (the state log of the reverse-mode kernels: GBs)
            t_g = time.perf_counter() - t0
        except NotImplementedError as e:
            unsupported = str(e)
        except _lib.ExtrackError as e:
            if e.code != _lib.E_UNSUPPORTED:
                raise
            unsupported = str(e)
        if comm is not None:  # every rank must take the same branch below
            if comm.allreduce_scalar(1.0 if unsupported else 0.0, "max") > 0.5:
                unsupported = unsupported or "not served on another rank"
        if unsupported is None:
            clear = t_g > 20.0 * (len(names) + 1) * t_ll  # hopelessly on the wrong side even with its allocations: spare the warm call
            if comm is not None:
                clear = comm.allreduce_scalar(1.0 if clear else 0.0, "min") > 0.5
            if not clear:
                t0 = time.perf_counter()
                cum_Proba_Cs_grad(params, names, *a)
                t_g = time.perf_counter() - t0
    if unsupported is not None:
        return decide(False, "the gradient kernels do not serve this model (%s)" % unsupported)
    if comm is not None:
        v = comm.allreduce_vector(np.array([t_ll, t_g]), op="max")
        t_ll, t_g = float(v[0]), float(v[1])
    use = t_g < (len(names) + 1) * t_ll
    return decide(use, "timing probe: one gradient call %.3g ms vs %d objective calls of %.3g ms" % (t_g * 1e3, len(names) + 1, t_ll * 1e3))


def _fit_threshold_frozen_plan(params, fargs, method, ts, max_rounds=6):
    """Threshold-fusion fit with the exact gradient.  The objective of extrack/tracking.py:991 re-decides its merge groups at every
    evaluation (fuse_tracks_th, :676-701), which makes it piecewise smooth: a quasi-Newton method fed with exact gradients of the pieces
    sees jumps its model cannot explain, needs several times the iterations of the fixed-window fit and ends on "precision loss".  So:
      1. evaluate once (the plan kernel decides the groups at the current parameters),
      2. FREEZE that plan and minimise the now smooth objective with its exact gradient (gradient kernel only, no plan kernel),
      3. re-plan at the minimiser; stop when the re-planned objective equals the frozen one (the plan did not change) or no longer improves.
    The result's ``residual`` is the reference's objective (own plan) at the returned parameters; ``nfev`` / ``ngev`` count every evaluation of
    every round; ``plan_rounds`` says how many plans were used."""
    from . import lmfit_compat
    nfev = ngev = 0
    cur, fit, f_prev, rounds, hinv = params, None, None, 0, None
    a = list(fargs)
    a[7] = 0  # the planning evaluations between the rounds are silent
    import contextlib
    import io
    try:
        for rounds in range(1, max_rounds + 1):
            ts.th_freeze_plan(False)
            with contextlib.redirect_stdout(io.StringIO()):
                f_plan = cum_Proba_Cs(cur, *a)  # decides the plan at `cur`
            nfev += 1
            if not np.isfinite(f_plan):
                raise ValueError("the starting parameters are invalid for the model (objective = inf)")
            ts.th_freeze_plan(True)
            # later rounds start at the previous round's minimiser: they also start from its inverse-Hessian estimate (scipy's BFGS takes one)
            # instead of re-learning the curvature from the identity
            kw = {}
            if hinv is not None and str(method).lower() == "bfgs":
                kw["options"] = {"hess_inv0": hinv}
            t_round = time.perf_counter()
            fit = lmfit_compat.minimize_with_gradient(cum_Proba_Cs, cur, args=fargs, method=method, nan_policy="propagate", fcn_grad=cum_Proba_Cs_grad, **kw)
            if os.environ.get("EXTRACK_FIT_TRACE"):
                sys.stderr.write("[fit] plan round %d: %d + %d calls, %.3f s, frozen objective %.6f\n" % (
                    rounds, int(fit.nfev), int(getattr(fit, "ngev", 0)), time.perf_counter() - t_round, float(fit.residual[0])))
            h = getattr(getattr(fit, "scipy_result", None), "hess_inv", None)
            hinv = None
            if isinstance(h, np.ndarray) and np.all(np.isfinite(h)):
                h = (np.asarray(h, float) + np.asarray(h, float).T) / 2
                w = np.linalg.eigvalsh(h)
                if w.min() > 1e-10 * w.max():  # scipy insists on a positive definite start; a round that ended on a failed line search may not leave one
                    hinv = h
            nfev += int(fit.nfev)
            ngev += int(getattr(fit, "ngev", 0))
            ts.th_freeze_plan(False)
            cur = getattr(fit, "own_params", fit.params)
            with contextlib.redirect_stdout(io.StringIO()):
                f_new = cum_Proba_Cs(cur, *a)  # the reference's objective (own plan) at the minimiser
            nfev += 1
            f_frozen = float(fit.residual[0])
            same_plan = abs(f_new - f_frozen) <= 1e-11 * abs(f_new)
            no_gain = f_prev is not None and f_new >= f_prev - 1e-10 * abs(f_prev)
            f_prev = f_new if f_prev is None else min(f_prev, f_new)
            if same_plan or no_gain:
                break
    finally:
        ts.th_freeze_plan(False)
    fit.residual = np.atleast_1d(np.float64(f_new))
    fit.nfev, fit.ngev, fit.plan_rounds = nfev, ngev, rounds
    if type(fit.params) is not type(params):  # the caller's Parameters type (real lmfit) carries the fitted values
        import copy
        out = copy.deepcopy(params)
        for k, p in cur.items():
            if not getattr(out[k], "expr", None):
                out[k].value = p.value
        if hasattr(out, "update_constraints"):
            out.update_constraints()
        fit.own_params, fit.params = cur, out
    return fit


# ------------------------------------------------------------------------------------------------------------
# public API
# ------------------------------------------------------------------------------------------------------------
def param_fitting(all_tracks, dt, params=None, nb_states=2, nb_substeps=1, frame_len=6, verbose=1, workers=1, Matrix_type=1,
                  method="bfgs", steady_state=False, cell_dims=[1], input_LocErr=None, threshold=0.2, max_nb_states=120,
                  device=None, comm=None, fusion=None, gradient=None):
    """Fit the model parameters to a length-bucketed track dict (extrack/tracking.py:1299-1386).

    all_tracks: {str(len): ndarray[n_tracks, len, dims]}.  Returns the lmfit (or lmfit_compat) MinimizerResult:
    ``.params[name].value``, ``.residual[0] == -log-likelihood``.  Extra keywords: ``device`` (GPU index; default: the
    communicator's device, else 0), ``comm`` (distributed shard communicator: every rank passes the WHOLE dataset and keeps its
    shard), ``fusion`` ("window" | "threshold" | None = process default, see the module docstring), ``gradient``
    ("analytic": the optimiser gets the exact gradient from the GPU, one evaluation per iteration; "fd": it differences the
    objective like the reference's; None: analytic where it exists (fixed-window kernel, gradient-based method, differentiable
    constraint expressions) AND a timing probe on this dataset says it is the cheaper way to a gradient, else fd: ``_pick_gradient``)."""
    fusion = "threshold" if _check_fusion(fusion) else "window"
    device = _resolve_device(device, comm)
    if params is None:
        params = generate_params(nb_states=nb_states, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3,
                                 Fractions_bounds=[0.001, 0.99], estimated_transition_rates=0.1)
    _, tracks, sigmas = engine.sort_buckets(all_tracks, input_LocErr)
    if len(tracks) < 1:
        raise ValueError("No track could be detected. The loaded tracks seem empty. Errors often come from wrong input paths.")
    if frame_len <= nb_substeps:  # tracking_0.py:1015-1017
        print("Warning frame_len has to be at least nb_substeps + 1")
        frame_len = nb_substeps + 1
    print("cell_dims", cell_dims)
    dt = _sorted_dt(dt, all_tracks)
    dts = dt if isinstance(dt, list) else None
    if dts is not None and fusion != "threshold":
        raise NotImplementedError("per-track time steps (dt as a dict of arrays) exist in the threshold-fusion kernel only "
                                  "(extrack/tracking.py:494-499): use fusion='threshold' (or EXTRACK_FUSION=threshold)")
    if comm is not None:  # per-track time steps are cut like the tracks (whole 2000-track chunks: a chunk's field-of-view table comes from ITS tracks)
        ts = comm.shard_trackset(tracks, sigmas, device=device, chunk=2000 if fusion == "threshold" else None, dts=dts)
    else:
        ts = TrackSet(tracks, sigmas, device=device, dts=dts)
    from . import lmfit_compat
    can_grad = str(method).lower() in lmfit_compat._GRADIENT_METHODS and not (fusion == "threshold" and dts is not None)
    if gradient not in (None, "analytic", "fd"):
        raise ValueError("gradient must be None, 'analytic' or 'fd'")
    if gradient == "analytic" and not can_grad:
        raise ValueError("gradient='analytic' needs a gradient-based method (and, with fusion='threshold', a scalar dt)")
    fargs = (ts, dt, cell_dims, sigmas, nb_states, nb_substeps, frame_len, verbose, workers, Matrix_type, threshold, max_nb_states, 2000,
             comm, fusion)
    ginfo = {"gradient_path": "fd", "gradient_why": "requested (gradient='fd')" if gradient == "fd" else "method %r takes no gradient" % method}
    try:
        use_grad = can_grad and gradient != "fd"
        if use_grad:
            use_grad = _pick_gradient(params, fargs, explicit=(gradient == "analytic"), comm=comm, info=ginfo)
        if use_grad and fusion == "threshold":
            fit = _fit_threshold_frozen_plan(params, fargs, method, ts)
        elif use_grad:
            # the built-in BFGS driver takes the analytic gradient (chain rule through the bounds transform applied there); with real
            # lmfit installed its Parameters are converted for the fit and the fitted values written back into a copy of them
            fit = lmfit_compat.minimize_with_gradient(cum_Proba_Cs, params, args=fargs, method=method, nan_policy="propagate",
                                                      fcn_grad=cum_Proba_Cs_grad)
        else:
            fit = minimize(cum_Proba_Cs, params, args=fargs, method=method, nan_policy="propagate")
    finally:
        ts.close()
    # which way the optimiser got its gradient, and why (the fit's record; `ngev` > 0 says the same for the built-in driver)
    fit.gradient_path, fit.gradient_why = ginfo["gradient_path"], ginfo["gradient_why"]
    if verbose == 0:
        print("")
    return fit


def predict_Bs(all_tracks, dt, params, cell_dims=[1], nb_states=4, frame_len=5, max_nb_states=200, threshold=0.1, workers=1,
               input_LocErr=None, verbose=0, nb_max=1, device=None, comm=None, fusion=None):
    """Probability of each localisation to be in each state (extrack/tracking.py:792-906).

    Returns {str(len): ndarray[n_tracks, len, nb_states]} keyed by every input key (empty arrays for empty
    buckets), rows in input order.  ``nb_substeps`` is forced to 1 like the reference (:839); min/max length
    come from ALL keys (:853-854).  With ``comm`` (extrack_amd.distributed.Comm) every rank annotates its row range of
    every bucket on its own GPU and rank 0 gets the row-ordered result (other ranks get None); no collective is needed in
    the data path."""
    fusion = "threshold" if _check_fusion(fusion) else "window"
    device = _resolve_device(device, comm)
    if comm is not None:
        from .distributed import shard_range
        # threshold fusion takes its merge decisions per chunk of nb_max consecutive tracks of a bucket (extrack/tracking.py:856-875): the
        # shards are whole chunks, so every rank annotates exactly the chunks a single GPU would
        ch = int(nb_max) if (_check_fusion(fusion) and nb_max and nb_max > 1) else None
        cut = lambda v: np.asarray(v)[slice(*shard_range(len(v), comm.rank, comm.world, ch))]
        loc_tracks = {k: cut(v) for k, v in all_tracks.items()}
        loc_sig = None if input_LocErr is None else {k: cut(v) for k, v in input_LocErr.items()}
        loc_dt = dt if not isinstance(dt, dict) else {k: cut(v) for k, v in dt.items()}
        local = predict_Bs(loc_tracks, loc_dt, params, cell_dims, nb_states, frame_len, max_nb_states, threshold, workers, loc_sig, verbose,
                           nb_max, device, None, fusion)
        return comm.gather_rows(local)
    keys, tracks, sigmas = engine.sort_buckets(all_tracks, input_LocErr)
    if not is_parameters(params):
        raise TypeError("params must be either of the class 'lmfit.parameter.Parameters' or a dictionary of the relevant parameters")
    nb_substeps = 1
    dt = _sorted_dt(dt, all_tracks)
    dts = dt if isinstance(dt, list) else None
    if dts is not None and fusion != "threshold":
        raise NotImplementedError("per-track time steps (dt as a dict of arrays) exist in the threshold-fusion kernel only "
                                  "(extrack/tracking.py:494-499): use fusion='threshold' (or EXTRACK_FUSION=threshold)")
    le, Ds, Fs, TrMat, pBL, so = _extract_arrays(params, dt, nb_substeps, 1)
    ds = np.sqrt(2 * Ds) if dts is not None else np.sqrt(2 * Ds * dt)
    S = len(ds)
    out = {l: np.empty((0, int(l), S)) for l in keys}
    if not tracks:
        return out
    ts = TrackSet(tracks, sigmas, device=device, min_len=max(int(keys[0]), 2), max_len=int(keys[-1]), dts=dts)
    try:
        if sigmas is not None:
            model = ts.make_model(None, ds, Fs, TrMat, pBL, cell_dims, 1, frame_len, slope_offset=so, dt_chunk=nb_max)
        else:
            model = ts.make_model(le[None, None], ds, Fs, TrMat, pBL, cell_dims, 1, frame_len, dt_chunk=nb_max)
        res = ts.predict_th(model, threshold, max_nb_states, nb_max) if _check_fusion(fusion) else ts.predict(model)
        for arr, pr in zip(tracks, res):
            out[str(arr.shape[1])] = pr
            if verbose:
                print(".", end="")
    finally:
        ts.close()
    return out
