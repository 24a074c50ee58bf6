"""ORACLE (test infrastructure, not product code) -- numpy restatement of the
ExTrack fixed-window track-likelihood recursion.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``extrack_amd`` never does.

What is restated (reference file:line, all relative to /root/reference/):
  * ``extrack/tracking.py:746-757``   get_all_Bs          -> implicit digits ``(i // S**c) % S``
  * ``extrack/tracking.py:759-767``   get_Ts_from_Bs      -> ``seq_tables`` (LTs)
  * ``extrack/tracking.py:174-180``   step variance       -> ``seq_tables`` (d2s)
  * ``extrack/tracking.py:182-192``   FOV/bleach table    -> ``p_stay_table``
  * ``extrack/tracking.py:76-107``    Gaussian-integral   -> inside ``p_cs_inter_bound_stats``
  * ``extrack/tracking.py:361-423``   fuse_tracks_general -> ``_fuse_oldest``
  * ``extrack/tracking.py:109-318``   P_Cs_inter_bound_stats (== tracking_0.py:96-305)
  * ``extrack/tracking_0.py:440-458`` Proba_Cs            -> ``proba_cs``
  * ``extrack/tracking_0.py:637-715`` cum_Proba_Cs        -> ``cum_proba_cs``
  * ``extrack/tracking_0.py:463-563`` predict_Bs          -> ``predict_bs``
  * ``extrack/tracking.py:913-986``   extract_params      -> ``extract_params``

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this module
against golden vectors produced by importing the reference itself
(``tests/golden/make_golden.py``), including the RNG-free known answers of
SURVEY.md Appendix B.

Deliberate deviation (documented, corner case only): the reference guards
``exp`` overflow with a *chunk-wide* shift ``max(LP) - 600``
(tracking.py:264-265, 309-310); here the shift is per track (row max), which
is identical unless a track's weights underflow relative to the chunk maximum
(the reference then returns NaN posteriors for that track).

The track is walked in original time order; the reference reverses the arrays
and walks backwards, which is the same thing (tracking.py:134).
Sequence index convention (identical to the reference): digit ``c`` of index
``i`` is ``(i // S**c) % S`` with c = 0 the NEWEST state.
"""
import numpy as np
from scipy.special import ndtr  # == scipy.stats.norm.cdf without the distribution machinery

LOG2PI = np.log(2 * np.pi)


def seq_tables(S, ns, ds, TrMat):
    """Tables over the ns+1 newest digits of a sequence index.

    LTs[j] = sum_c log T[g_{c+1}(j) -> g_c(j)]          (tracking.py:759-767, TrMat.T at :155)
    d2s[j] = mean_c (ds[g_c]^2 + ds[g_{c+1}]^2) / 2     (tracking.py:174-180)
    """
    j = np.arange(S ** (ns + 1))
    dig = [(j // S ** c) % S for c in range(ns + 1)]
    logT = np.log(np.asarray(TrMat, float))
    ds2 = np.asarray(ds, float) ** 2
    LTs = np.zeros(len(j))
    d2s = np.zeros(len(j))
    for c in range(ns):
        LTs += logT[dig[c + 1], dig[c]]
        d2s += (ds2[dig[c]] + ds2[dig[c + 1]]) / 2
    return LTs, d2s / ns


def p_stay_table(ds, S, ns, cell_dims):
    """Probability of staying in the field of view for each of the S**ns newest sub-sequences
    (tracking.py:182-191).  Index r has digits (r // S**c) % S, c < ns."""
    r = np.arange(S ** ns)
    ds2 = np.asarray(ds, float) ** 2
    sub = np.zeros(len(r))
    for c in range(ns):
        sub += ds2[(r // S ** c) % S]
    sub_ds = (sub / ns) ** 0.5
    p_stay = np.ones(len(r))
    for cell_len in cell_dims:
        xs = np.linspace(0 + cell_len / 2000, cell_len - cell_len / 2000, 1000)
        cur = np.mean(ndtr((cell_len - xs[:, None]) / (sub_ds + 1e-200)) - ndtr(-xs[:, None] / (sub_ds + 1e-200)), 0)
        p_stay = p_stay * cur
    return p_stay


def _fuse_oldest(LP, m, s2, S):
    """Softmax-weighted merge over the oldest (most significant) digit (tracking.py:361-423)."""
    N, nB = LP.shape
    LPr = LP.reshape(N, S, nB // S)
    mx = LPr.max(axis=1, keepdims=True)
    w = np.exp(LPr - mx)
    sw = w.sum(axis=1, keepdims=True)
    a = (w / sw)[..., None]
    m = (a * m.reshape(N, S, nB // S, m.shape[-1])).sum(axis=1)
    s2 = (a * s2.reshape(N, S, nB // S, s2.shape[-1])).sum(axis=1)
    LP = np.log(sw[:, 0]) + mx[:, 0]
    return LP, m, s2


def p_cs_inter_bound_stats(Cs, LocErr, ds, Fs, TrMat, pBL=0.1, isBL=1, cell_dims=(0.5,), nb_substeps=1,
                           frame_len=4, do_preds=0, min_len=3):
    """Returns (LP[N, nB_final], preds[N, L, S] or None).  tracking.py:109-318."""
    Cs = np.asarray(Cs, float)
    N, L, D = Cs.shape
    ds = np.asarray(ds, float)
    Fs = np.asarray(Fs, float)
    TrMat = np.asarray(TrMat, float)
    S = TrMat.shape[0]
    ns, F = int(nb_substeps), int(frame_len)
    LocErr = np.asarray(LocErr, float)
    if LocErr.ndim != 3 or LocErr.shape[1] not in (1, L):
        raise ValueError("Localization error is not specified correctly")  # tracking.py:143
    if LocErr.shape[1] == 1 and L != 1:
        l2 = lambda p: LocErr[:, 0, :] ** 2
    else:
        l2 = lambda p: LocErr[:, p, :] ** 2
    if L < 2:
        raise ValueError("minimal track length = 2, here track length = %s" % L)  # tracking.py:150
    if do_preds and ns != 1:
        raise ValueError("state predictions require nb_substeps == 1")  # the reference raises IndexError here
    k = LocErr.shape[2]
    G = S ** ns
    LTs, d2s = seq_tables(S, ns, ds, TrMat)
    pst = p_stay_table(ds, S, ns, cell_dims)
    Lpst = np.log(pst * (1 - pBL))
    preds = np.zeros((N, L, S)) - 1 if do_preds else None

    def gauss_log(c, m, s2x, half):
        # sum_d ( -half*log(2 pi s2x_d) - (c_d - m_d)^2 / (2 s2x_d) ), a k=1 variance broadcasts over D dims
        return np.sum(-half * np.log(2 * np.pi * s2x) - (c - m) ** 2 / (2 * s2x), axis=2)

    # step 1: first position (tracking.py:152-198)
    n = ns + 1
    idx = np.arange(S ** n)
    LP = np.repeat((LTs[idx] + np.log(Fs[(idx // S ** (n - 1)) % S]))[None], N, axis=0)
    m = np.repeat(Cs[:, 0, None, :], S ** n, axis=1)
    s2 = np.broadcast_to(l2(0)[:, None, :] + d2s[idx][None, :, None], (N, S ** n, k)).copy()

    for t in range(2, L):  # tracking.py:210-280, injects position p = t-1
        p = t - 1
        n += ns
        idx = np.arange(S ** n)
        par = idx // G
        sm = idx % S ** (ns + 1)
        lp = l2(p)[:, None, :]
        c = Cs[:, p, None, :]
        mo, s2o = m[:, par], s2[:, par]
        den = lp + s2o
        if k == 1:  # tracking.py:94-95
            LC = D * -0.5 * np.log(2 * np.pi * den[:, :, 0]) - np.sum((c - mo) ** 2 / (2 * den), axis=2)
        else:       # tracking.py:97
            LC = np.sum(-0.5 * np.log(2 * np.pi * den), 2) - np.sum((c - mo) ** 2 / (2 * den), axis=2)
        m = (mo * lp + c * s2o) / den
        d2e = d2s[sm][None, :, None]
        s2 = (d2e * lp + d2e * s2o + lp * s2o) / den
        LP = LP[:, par] + LTs[sm][None] + LC
        if t >= min_len:
            LP = LP + Lpst[idx % G][None]
        if t < L - 1:  # tracking.py:253
            while n > F:
                if do_preds:  # tracking.py:255-271 (note: no 1/2 on the log -- reference quirk)
                    tl = LP + gauss_log(Cs[:, p + 1, None, :], m, s2 + l2(p + 1)[:, None, :], 1.0)
                    P = np.exp(tl - tl.max(axis=1, keepdims=True))
                    preds[:, t - F, :] = P.reshape(N, S, -1).sum(axis=2) / P.sum(axis=1, keepdims=True)
                LP, m, s2 = _fuse_oldest(LP, m, s2, S)
                n -= 1

    if isBL:  # tracking.py:282-299
        n += ns
        idx = np.arange(S ** n)
        par = idx // G
        sm = idx % S ** (ns + 1)
        end_p = pst[idx % S]  # p_stay indexed by the raw newest state value (tracking.py:297)
        LL = np.log(pBL + (1 - end_p) - pBL * (1 - end_p)) + LTs[sm]
        LP, m, s2 = LP[:, par], m[:, par], s2[:, par]
    else:
        idx = np.arange(S ** n)
        LL = 0.0
    LP = LP + gauss_log(Cs[:, L - 1, None, :], m, s2 + l2(L - 1)[:, None, :], 0.5) + LL  # tracking.py:301-306

    if do_preds:  # tracking.py:308-317
        P = np.exp(LP - LP.max(axis=1, keepdims=True))
        tot = P.sum(axis=1)
        for col in range(int(isBL), n):
            dig = (idx // S ** col) % S
            for s in range(S):
                preds[:, (L - 1) - (col - int(isBL)), s] = P[:, dig == s].sum(axis=1) / tot
    return LP, preds


def proba_cs(Cs, LocErr, ds, Fs, TrMat, pBL, isBL, cell_dims, nb_substeps, frame_len, min_len):
    """Per-track log-likelihood LP_C[N] (tracking_0.py:440-458)."""
    LP, _ = p_cs_inter_bound_stats(Cs, LocErr, ds, Fs, TrMat, pBL, isBL, cell_dims, nb_substeps, frame_len, 0, min_len)
    mx = LP.max(axis=1, keepdims=True)
    return np.log(np.exp(LP - mx).sum(axis=1)) + mx[:, 0]


def extract_params(values, dt, nb_substeps=1, Matrix_type=1):
    """values: {name: float}.  tracking.py:913-986 for the default Matrix_type (0/1) and a global LocErr."""
    names = np.sort(list(values.keys()))
    LocErr = np.array([values[n] for n in names if n.startswith("LocErr")], float)[None, None]
    Ds = np.array([values[n] for n in names if n.startswith("D") and len(n) < 3], float)
    Fs = np.array([values[n] for n in names if n.startswith("F")], float)
    S = len(Ds)
    TrMat = np.zeros((S, S))
    pBL = None
    for n in values:
        if n == "pBL":
            pBL = values[n]
        elif n.startswith("p"):
            TrMat[int(n[1]), int(n[2])] = values[n]
    TrMat = TrMat / nb_substeps
    if Matrix_type == 1:
        TrMat = 1 - np.exp(-TrMat)
    elif Matrix_type != 0:
        raise NotImplementedError("oracle restates Matrix_type 0 and 1 only")
    TrMat[np.arange(S), np.arange(S)] = 0
    TrMat[np.arange(S), np.arange(S)] = 1 - TrMat.sum(1)
    return LocErr, np.sqrt(2 * Ds * dt), Fs, TrMat, pBL


def _sorted_buckets(all_tracks, input_LocErr=None):
    keys = np.sort(np.array(list(all_tracks.keys())).astype(int)).astype(str)
    tr, le = [], []
    for l in keys:
        if len(all_tracks[l]) > 0:
            tr.append(np.asarray(all_tracks[l], float))
            if input_LocErr is not None:
                le.append(np.asarray(input_LocErr[l], float))
    return keys, tr, le


def cum_proba_cs(values, all_tracks, dt, cell_dims=(1,), input_LocErr=None, nb_substeps=1, frame_len=6,
                 Matrix_type=1, chunk=2000, per_track=False):
    """-sum(LL) over a length-bucketed dict (tracking_0.py:637-715; chunk 50 there, 2000 in tracking.py:991).
    ``values`` is a plain {name: float} dict.  Returns +inf for invalid parameters."""
    LocErr, ds, Fs, TrMat, pBL = extract_params(values, dt, nb_substeps, Matrix_type)
    _, buckets, les = _sorted_buckets(all_tracks, input_LocErr)
    min_len, max_len = buckets[0].shape[1], buckets[-1].shape[1]
    if not (np.all(TrMat > 0) and np.all(Fs > 0) and np.all(ds[1:] - ds[:-1] >= 0)):
        return np.inf
    out = []
    for b, Css in enumerate(buckets):
        isBL = 0 if Css.shape[1] == max_len else 1
        for a in range(0, len(Css), chunk):
            le = LocErr if input_LocErr is None else les[b][a:a + chunk]
            out.append(proba_cs(Css[a:a + chunk], le, ds, Fs, TrMat, pBL, isBL, cell_dims, nb_substeps, frame_len, min_len))
    out = np.concatenate(out)
    if per_track:
        return out
    tot = -np.sum(out)
    return np.inf if np.isnan(tot) else tot


def predict_bs(values, all_tracks, dt, cell_dims=(1,), frame_len=8, input_LocErr=None, chunk=50):
    """State posteriors per position (tracking_0.py:463-563): nb_substeps forced to 1, min/max length from all keys."""
    LocErr, ds, Fs, TrMat, pBL = extract_params(values, dt, 1, 1)
    keys, buckets, les = _sorted_buckets(all_tracks, input_LocErr)
    min_len, max_len = int(keys[0]), int(keys[-1])
    S = len(ds)
    res = {l: np.empty((0, int(l), S)) for l in keys}
    for b, Css in enumerate(buckets):
        isBL = 0 if Css.shape[1] == max_len else 1
        parts = []
        for a in range(0, len(Css), chunk):
            le = LocErr if input_LocErr is None else les[b][a:a + chunk]
            parts.append(p_cs_inter_bound_stats(Css[a:a + chunk], le, ds, Fs, TrMat, pBL, isBL, cell_dims, 1, frame_len, 1, min_len)[1])
        res[str(Css.shape[1])] = np.concatenate(parts)
    return res
