"""Device-resident track sets and the model-table plumbing between the Python API and the C ABI.

A ``TrackSet`` is the MI355X counterpart of the argument list that the reference rebuilds for
every objective evaluation (extrack/tracking.py:1019-1056): the length buckets are uploaded to HBM
ONCE and stay resident for the whole fit; an evaluation only ships the ~100-byte model tables.
"""
import numpy as np
from scipy.special import ndtr

from . import _lib


_P_STAY_CACHE = {}


def p_stay_table(ds, nb_states, nb_substeps, cell_dims):
    """Cached front of ``_p_stay_table``: during a fit most evaluations (finite-difference steps on the non-diffusion
    parameters) repeat the same diffusion lengths."""
    key = (tuple(np.asarray(ds, float).tolist()), int(nb_states), int(nb_substeps), tuple(float(c) for c in cell_dims))
    hit = _P_STAY_CACHE.get(key)
    if hit is None:
        if len(_P_STAY_CACHE) > 256:
            _P_STAY_CACHE.clear()
        hit = _P_STAY_CACHE[key] = _p_stay_table(ds, nb_states, nb_substeps, cell_dims)
    return hit


def _p_stay_table(ds, nb_states, nb_substeps, cell_dims):
    """Probability of staying in the field of view for each of the S**ns newest sub-sequences.

    Same quadrature as the reference (extrack/tracking.py:182-191): mean over a 1000-point grid of
    Phi((cell - x)/sigma) - Phi(-x/sigma), product over cell dimensions, with
    sigma_r = sqrt(mean_c ds[g_c(r)]**2) and digit c of r = (r // S**c) % S.
    """
    S, ns = int(nb_states), int(nb_substeps)
    ds2 = np.asarray(ds, float) ** 2
    r = np.arange(S ** ns)
    acc = np.zeros(len(r))
    for c in range(ns):
        acc += ds2[(r // S ** c) % S]
    sub_ds = np.sqrt(acc / ns)
    out = np.ones(len(r))
    for cell_len in cell_dims:
        xs = np.linspace(0 + cell_len / 2000, cell_len - cell_len / 2000, 1000)[:, None]
        out = out * np.mean(ndtr((cell_len - xs) / (sub_ds + 1e-200)) - ndtr(-xs / (sub_ds + 1e-200)), 0)
    return out


def p_stay_table_grad(ds, nb_states, nb_substeps, cell_dims):
    """(p_stay [G], d p_stay / d(ds^2) [G, S]) for the gradient path: the table of ``_p_stay_table`` and its analytic derivative
    with respect to the SQUARED diffusion lengths (finite at ds = 0, where the table is flat).
    With sigma_r = sqrt(mean_c ds^2[g_c(r)]) and one cell dimension c: p = mean_x [Phi((c - x)/sigma) - Phi(-x/sigma)],
    dp/dsigma = -mean_x [phi((c - x)/sigma) (c - x) + phi(x/sigma) x] / sigma^2, d sigma / d ds^2[s] = count_s(r) / (2 ns sigma)."""
    S, ns = int(nb_states), int(nb_substeps)
    ds2 = np.asarray(ds, float) ** 2
    r = np.arange(S ** ns)
    cnt = np.zeros((len(r), S))
    for c in range(ns):
        cnt[r, (r // S ** c) % S] += 1
    sig = np.sqrt(cnt @ ds2 / ns)
    p = np.ones(len(r))
    dlogp = np.zeros(len(r))  # d log p / d sigma, summed over the cell dimensions
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        for cell_len in cell_dims:
            xs = np.linspace(0 + cell_len / 2000, cell_len - cell_len / 2000, 1000)[:, None]
            a, b = (cell_len - xs) / (sig + 1e-200), xs / (sig + 1e-200)
            cur = np.mean(ndtr(a) - ndtr(-b), 0)
            phi = lambda z: np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi)
            num = np.mean(phi(a) * a + phi(b) * b, 0)  # = mean[phi(.)(c-x) + phi(.)x] / sigma
            dcur = np.where(num > 0, -num / (sig + 1e-200), 0.0)
            p = p * cur
            dlogp = dlogp + np.where(cur > 0, dcur / np.where(cur > 0, cur, 1.0), 0.0)
        dsig = np.where(sig[:, None] > 0, cnt / (2 * ns * np.where(sig > 0, sig, 1.0)[:, None]), 0.0)
    dp = np.nan_to_num((p * dlogp)[:, None] * dsig, nan=0.0, posinf=0.0, neginf=0.0)
    return p, dp


_P_STAY_GRAD_CACHE = {}


def p_stay_table_grad_cached(ds, nb_states, nb_substeps, cell_dims):
    """Cached front of ``p_stay_table_grad`` (an analytic-gradient fit asks for the same diffusion lengths it evaluates at)."""
    key = (tuple(np.asarray(ds, float).tolist()), int(nb_states), int(nb_substeps), tuple(float(c) for c in cell_dims))
    hit = _P_STAY_GRAD_CACHE.get(key)
    if hit is None:
        if len(_P_STAY_GRAD_CACHE) > 64:
            _P_STAY_GRAD_CACHE.clear()
        hit = _P_STAY_GRAD_CACHE[key] = p_stay_table_grad(ds, nb_states, nb_substeps, cell_dims)
    return hit


def sort_buckets(all_tracks, input_LocErr=None):
    """Numeric sort of the length keys, dropping empty buckets (extrack/tracking.py:1346-1367).

    Returns (all_keys_sorted, [track arrays short->long], [sigma arrays] or None)."""
    keys = np.sort(np.array(list(all_tracks.keys())).astype(int)).astype(str)
    tracks, sigmas = [], []
    for l in keys:
        if len(all_tracks[l]) > 0:
            tracks.append(np.asarray(all_tracks[l], dtype=np.float64))
            if input_LocErr is not None:
                sigmas.append(np.asarray(input_LocErr[l], dtype=np.float64))
    return list(keys), tracks, (sigmas if input_LocErr is not None else None)


class TrackSet:
    """Length buckets resident on one GPU.

    buckets: list of arrays [N_l, l, D] sorted short -> long (as the reference's objective receives them).
    sigmas:  optional list of per-peak localisation errors with matching shapes [N_l, l, 1|D].
    min_len / max_len: dataset-global values (default: from ``buckets``); a shard of a distributed
    dataset must be given the global ones (extrack/tracking.py:1009-1010 uses the whole list).
    """

    def __init__(self, buckets, sigmas=None, device=0, min_len=None, max_len=None, allow_empty=False, dts=None):
        """allow_empty: a shard of a distributed dataset may hold no track at all (its objective is 0.0, its posteriors are
        empty); the dataset-global ``min_len`` / ``max_len`` must then be given.
        dts: optional list of per-track time steps [N_l, l] matching ``buckets`` (extrack/tracking.py:979-982: ``dt`` given as a
        dict of arrays); only the threshold-fusion kernels take them."""
        if len(buckets) < 1 and not allow_empty:
            raise ValueError("No track could be detected. The loaded tracks seem empty. Errors often come from wrong input paths.")
        self.ctx = _lib.Context(device)
        self.shapes = []
        self.dt0 = []  # first column of every bucket's dt array (host copy): what the per-chunk field-of-view tables are made from
        self.has_dt = dts is not None
        for i, b in enumerate(buckets):
            b = np.asarray(b, dtype=np.float64)
            if b.ndim != 3:
                raise ValueError("each bucket must be an array [n_tracks, len, dims]")
            if b.shape[1] < 2:
                raise ValueError("minimal track length = 2, here track length = %s" % b.shape[1])
            s = None if sigmas is None else np.asarray(sigmas[i], dtype=np.float64)
            if s is not None and (s.ndim != 3 or s.shape[:2] != b.shape[:2] or s.shape[2] not in (1, b.shape[2])):
                raise ValueError("Localization error is not specified correctly: input_LocErr must match all_tracks")
            if len(b):
                bid = self.ctx.upload_bucket(b, s)
                self.shapes.append(b.shape)
                if dts is not None:
                    t = np.asarray(dts[i], dtype=np.float64)
                    if t.shape != b.shape[:2]:
                        raise ValueError("dt must be a float or, per bucket, an array [n_tracks, len] matching all_tracks")
                    self.ctx.set_bucket_dt(bid, t)
                    self.dt0.append(np.array(t[:, 0]))
        if not self.shapes and not (allow_empty and min_len is not None and max_len is not None):
            self.ctx.close()
            raise ValueError("No track could be detected. The loaded tracks seem empty. Errors often come from wrong input paths.")
        lens = [s[1] for s in self.shapes]
        self.min_len = int(min(lens) if min_len is None else min_len)
        self.max_len = int(max(lens) if max_len is None else max_len)
        self.has_sigma = sigmas is not None
        self.n_tracks = int(sum(s[0] for s in self.shapes))
        self.dims = int(self.shapes[0][2]) if self.shapes else 0

    # ---- model handle -------------------------------------------------------------------------------------
    def make_model(self, LocErr, ds, Fs, TrMat, pBL, cell_dims, nb_substeps, frame_len, slope_offset=None, dt_chunk=None):
        """LocErr: global localisation error array of shape (1,1,k) (k = 1 or dims), or None when the per-peak
        errors uploaded with the buckets are to be used (then slope_offset = (slope, offset) or None).
        With per-track time steps (``dts``): ``ds`` are the diffusion lengths of a UNIT time step, sqrt(2 D), and ``dt_chunk`` the
        chunk size of the evaluation the model is for - the field-of-view table of a chunk is computed from the median over its
        tracks of sqrt(2 D dt[track, 0]) (extrack/tracking.py:507-511)."""
        ds = np.asarray(ds, float)
        S = len(ds)
        if self.has_dt:
            if not dt_chunk:
                raise ValueError("per-track time steps: the model needs the chunk size of the evaluation (dt_chunk)")
            tabs = []
            for t0 in self.dt0:
                for a in range(0, len(t0), int(dt_chunk)):
                    med = np.median(np.sqrt(ds[None] ** 2 * t0[a:a + int(dt_chunk), None]), axis=0)
                    tabs.append(p_stay_table(med, S, nb_substeps, cell_dims))
            if not tabs:  # a rank without tracks (distributed shard): one placeholder table, never indexed
                tabs.append(p_stay_table(ds, S, nb_substeps, cell_dims))
            ps = np.array(tabs)
        else:
            ps = p_stay_table(ds, S, nb_substeps, cell_dims)
        if LocErr is None:
            if not self.has_sigma:
                raise ValueError("per-peak localisation errors requested but none were uploaded")
            mode, le = (2, None) if slope_offset is not None else (1, None)
            slope, offset = slope_offset if slope_offset is not None else (0.0, 0.0)
        else:
            le = np.asarray(LocErr, float).reshape(-1)
            if len(le) not in (1, self.dims) and self.shapes:
                raise ValueError("Localization error is not specified correctly, in case of unique localization error specify a float "
                                 "number; if one localization error per dimension, specify one value per dimension")
            mode, slope, offset = 0, 0.0, 0.0
        return _lib.ModelHandle(ds, Fs, TrMat, ps, pBL, nb_substeps, frame_len, self.min_len, self.max_len, locerr=le,
                                locerr_mode=mode, slope=slope, offset=offset)

    def loglik(self, model, per_track=False):
        if not self.shapes:
            return (0.0, np.empty(0)) if per_track else 0.0
        return self.ctx.loglik(model, per_track=per_track)

    def loglik_th(self, model, threshold=0.2, max_nb_states=120, chunk=2000, per_track=False):
        if not self.shapes:
            return (0.0, np.empty(0)) if per_track else 0.0
        return self.ctx.loglik_th(model, threshold, max_nb_states, chunk, per_track=per_track)

    def th_freeze_plan(self, on):
        """Threshold-fusion evaluations follow the plan of the last planning evaluation (True) or decide their own again (False);
        see extrack_th_freeze_plan in include/extrack_hip.h.  A shard without tracks has nothing to freeze."""
        if self.shapes:
            self.ctx.th_freeze_plan(on)

    def predict_th(self, model, threshold=0.1, max_nb_states=200, nb_max=1):
        return [self.ctx.predict_th(model, i, threshold, max_nb_states, nb_max) for i in range(len(self.shapes))]

    def predict(self, model):
        """Posteriors for every uploaded bucket, in upload order: list of arrays [N_l, l, S]."""
        return [self.ctx.predict(model, i) for i in range(len(self.shapes))]

    def close(self):
        self.ctx.close()
