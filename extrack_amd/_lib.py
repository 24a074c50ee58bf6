"""ctypes binding of libextrack_hip.so (C ABI declared in include/extrack_hip.h).

There is deliberately NO CPU fallback: if the shared library or a gfx950 device is missing the
functions here raise, loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXTRACK_HIP_LIB", os.path.join(_HERE, "libextrack_hip.so"))

EXPORTS = [
    "extrack_abi_version", "extrack_create", "extrack_destroy", "extrack_last_error", "extrack_set_stream",
    "extrack_upload_bucket", "extrack_attach_bucket", "extrack_set_bucket_dt", "extrack_clear_buckets", "extrack_bucket_count",
    "extrack_loglik", "extrack_loglik_async", "extrack_predict", "extrack_last_kernel_ms",
    "extrack_last_launch_info", "extrack_p_stay_table", "extrack_loglik_th", "extrack_loglik_th_async", "extrack_th_plan_step",
    "extrack_predict_th", "extrack_loglik_grad", "extrack_loglik_grad_async", "extrack_last_grad_ms", "extrack_segment_len_hist", "extrack_refine_positions",
    "extrack_refine_pos_pdf",
    "extrack_sequence_columns", "extrack_sequence_matrix", "extrack_loglik_th_grad", "extrack_loglik_th_grad_async", "extrack_th_freeze_plan", "extrack_sequence_matrix_th",
    "extrack_multi_create", "extrack_multi_destroy", "extrack_multi_last_error", "extrack_multi_device_count", "extrack_multi_uses_rccl",
    "extrack_multi_context", "extrack_multi_upload_bucket", "extrack_multi_clear_buckets", "extrack_multi_loglik",
]

_dp = C.POINTER(C.c_double)


class ExtrackModel(C.Structure):
    """Mirror of ``struct extrack_model`` (include/extrack_hip.h)."""
    _fields_ = [
        ("n_states", C.c_int32), ("nb_substeps", C.c_int32), ("frame_len", C.c_int32), ("min_len", C.c_int32),
        ("max_len", C.c_int32), ("locerr_mode", C.c_int32), ("locerr_dims", C.c_int32), ("n_p_stay", C.c_int32),
        ("locerr", C.c_double * 3), ("slope", C.c_double), ("offset", C.c_double), ("pBL", C.c_double),
        ("ds", _dp), ("Fs", _dp), ("TrMat", _dp), ("p_stay", _dp),
    ]


class ExtrackModelTangent(C.Structure):
    """Mirror of ``struct extrack_model_tangent`` (include/extrack_hip.h)."""
    _fields_ = [
        ("locerr", C.c_double * 3), ("slope", C.c_double), ("offset", C.c_double), ("pBL", C.c_double),
        ("ds2", _dp), ("Fs", _dp), ("TrMat", _dp), ("p_stay", _dp),
    ]


E_INVALID, E_HIP, E_NODEVICE, E_UNSUPPORTED = -1, -2, -3, -4  # include/extrack_hip.h


class ExtrackError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libextrack_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (soname
    ``libamdhip64.so.7``) and link it by the unversioned name; if this extension pulled in the system copy first, a later
    ``import torch`` would load a SECOND runtime that then finds no GPU.  Preloading torch's copy (when torch is installed,
    without importing it) makes the extension's ``NEEDED libamdhip64.so.7`` resolve to that same object."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Loads the shared library; raises if it has not been built (``python __graft_entry__.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.extrack_abi_version.restype = C.c_int
    lib.extrack_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.extrack_destroy.argtypes = [vp]
    lib.extrack_destroy.restype = None
    lib.extrack_last_error.argtypes = [vp]
    lib.extrack_last_error.restype = C.c_char_p
    lib.extrack_set_stream.argtypes = [vp, vp]
    lib.extrack_upload_bucket.argtypes = [vp, vp, i64, i32, i32, vp, i32, C.POINTER(i32)]
    lib.extrack_attach_bucket.argtypes = [vp, vp, i64, i32, i32, vp, i32, C.POINTER(i32)]
    lib.extrack_set_bucket_dt.argtypes = [vp, i32, vp]
    lib.extrack_clear_buckets.argtypes = [vp]
    lib.extrack_bucket_count.argtypes = [vp]
    lib.extrack_loglik.argtypes = [vp, C.POINTER(ExtrackModel), _dp, vp]
    lib.extrack_loglik_async.argtypes = [vp, C.POINTER(ExtrackModel), vp]
    lib.extrack_predict.argtypes = [vp, C.POINTER(ExtrackModel), i32, vp]
    lib.extrack_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.extrack_last_launch_info.argtypes = [vp, C.POINTER(i32 * 6)]
    lib.extrack_p_stay_table.argtypes = [vp, i32, i32, vp, i32, vp]
    lib.extrack_loglik_th.argtypes = [vp, C.POINTER(ExtrackModel), C.c_double, i32, i32, _dp, vp]
    lib.extrack_loglik_th_async.argtypes = [vp, C.POINTER(ExtrackModel), C.c_double, i32, i32, vp]
    lib.extrack_predict_th.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.c_double, i32, i32, vp]
    lib.extrack_th_plan_step.argtypes = [vp, i32, i64, i32, C.POINTER(i32), C.POINTER(i32), vp, vp, i32]
    lib.extrack_loglik_grad.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.POINTER(ExtrackModelTangent), _dp, vp]
    lib.extrack_loglik_grad_async.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.POINTER(ExtrackModelTangent), vp]
    lib.extrack_segment_len_hist.argtypes = [vp, C.POINTER(ExtrackModel), i32, i32, vp]
    lib.extrack_refine_positions.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.c_double, i32, vp, vp]
    lib.extrack_refine_pos_pdf.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.c_double, i32, vp, i64, vp, vp, vp]
    lib.extrack_last_grad_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.extrack_sequence_columns.argtypes = [i32, i32, i32, i32, i32]
    lib.extrack_sequence_columns.restype = i64
    lib.extrack_sequence_matrix.argtypes = [vp, C.POINTER(ExtrackModel), i32, vp, i64]
    lib.extrack_loglik_th_grad.argtypes = [vp, C.POINTER(ExtrackModel), C.c_double, i32, i32, i32, C.POINTER(ExtrackModelTangent), _dp, vp]
    lib.extrack_loglik_th_grad_async.argtypes = [vp, C.POINTER(ExtrackModel), C.c_double, i32, i32, i32, C.POINTER(ExtrackModelTangent), vp]
    lib.extrack_th_freeze_plan.argtypes = [vp, i32]
    lib.extrack_sequence_matrix_th.argtypes = [vp, C.POINTER(ExtrackModel), i32, C.c_double, i32, vp, i64, C.POINTER(i64)]
    lib.extrack_multi_create.argtypes = [i32, C.POINTER(i32), i32, C.POINTER(vp)]
    lib.extrack_multi_destroy.argtypes = [vp]
    lib.extrack_multi_destroy.restype = None
    lib.extrack_multi_last_error.argtypes = [vp]
    lib.extrack_multi_last_error.restype = C.c_char_p
    lib.extrack_multi_device_count.argtypes = [vp]
    lib.extrack_multi_uses_rccl.argtypes = [vp]
    lib.extrack_multi_context.argtypes = [vp, i32]
    lib.extrack_multi_context.restype = vp
    lib.extrack_multi_upload_bucket.argtypes = [vp, vp, i64, i32, i32, vp, i32]
    lib.extrack_multi_clear_buckets.argtypes = [vp]
    lib.extrack_multi_loglik.argtypes = [vp, C.POINTER(ExtrackModel), _dp]
    if lib.extrack_abi_version() != 6:
        raise ImportError("libextrack_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class ModelHandle:
    """Keeps the numpy arrays referenced by an ExtrackModel alive."""

    def __init__(self, ds, Fs, TrMat, p_stay, pBL, nb_substeps, frame_len, min_len, max_len, locerr=None, locerr_mode=0,
                 slope=0.0, offset=0.0):
        self.ds, self.Fs, self.TrMat, self.p_stay = _f64(ds), _f64(Fs), _f64(TrMat), _f64(p_stay)
        S = len(self.ds)
        G = S ** int(nb_substeps)
        ok_ps = self.p_stay.shape == (G,) or (self.p_stay.ndim == 2 and self.p_stay.shape[1] == G)
        if self.TrMat.shape != (S, S) or self.Fs.shape != (S,) or not ok_ps:
            raise ValueError("inconsistent model array shapes")
        m = ExtrackModel()
        m.n_p_stay = 1 if self.p_stay.ndim == 1 else int(self.p_stay.shape[0])  # one table, or one per chunk (per-track time steps)
        m.n_states, m.nb_substeps, m.frame_len = S, int(nb_substeps), int(frame_len)
        m.min_len, m.max_len = int(min_len), int(max_len)
        m.locerr_mode = int(locerr_mode)
        le = np.zeros(3)
        if locerr_mode == 0:
            v = np.atleast_1d(np.asarray(locerr, float)).ravel()
            if len(v) < 1 or len(v) > 3:
                raise ValueError("global localisation error must have 1..3 components")
            le[:len(v)] = v
            m.locerr_dims = len(v)
        else:
            m.locerr_dims = 1
        m.locerr = (C.c_double * 3)(*le)
        m.slope, m.offset, m.pBL = float(slope), float(offset), float(pBL)
        m.ds = self.ds.ctypes.data_as(_dp)
        m.Fs = self.Fs.ctypes.data_as(_dp)
        m.TrMat = self.TrMat.ctypes.data_as(_dp)
        m.p_stay = self.p_stay.ctypes.data_as(_dp)
        self.c = m


class Context:
    """One libextrack_hip context = one GPU, one stream, a set of uploaded length buckets."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.extrack_create(int(device), C.byref(h))
        if rc != 0:
            raise ExtrackError(rc, self._lib.extrack_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.buckets = []  # (N, L, D, KS)
        self._keep = []    # attached device tensors

    def _check(self, rc):
        if rc != 0:
            raise ExtrackError(rc, self._lib.extrack_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.extrack_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self._lib.extrack_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def upload_bucket(self, tracks, sigma=None):
        tracks = _f64(tracks)
        if tracks.ndim != 3:
            raise ValueError("a bucket is an array [n_tracks, len, dims]")
        N, L, D = tracks.shape
        KS = 0
        sp = None
        if sigma is not None:
            sigma = _f64(sigma)
            if sigma.ndim != 3 or sigma.shape[:2] != (N, L):
                raise ValueError("Localization error is not specified correctly")
            KS = sigma.shape[2]
            sp = sigma.ctypes.data_as(C.c_void_p)
        bid = C.c_int32(-1)
        self._check(self._lib.extrack_upload_bucket(self._h, tracks.ctypes.data_as(C.c_void_p), N, L, D, sp, KS, C.byref(bid)))
        self.buckets.append((N, L, D, KS))
        return bid.value

    def set_bucket_dt(self, bucket_id, dt):
        """Per-track time steps [n_tracks, len] of an uploaded bucket (None removes them); see include/extrack_hip.h."""
        if dt is not None:
            dt = _f64(dt)
            N, L, D, KS = self.buckets[bucket_id]
            if dt.shape != (N, L):
                raise ValueError("dt must be an array [n_tracks, len] matching the bucket")
        self._check(self._lib.extrack_set_bucket_dt(self._h, int(bucket_id), dt.ctypes.data_as(C.c_void_p) if dt is not None else None))

    def attach_bucket(self, tracks_t, sigma_t=None):
        """tracks_t / sigma_t: contiguous float64 torch CUDA tensors on this device (zero copy)."""
        N, L, D = tracks_t.shape
        KS = 0 if sigma_t is None else sigma_t.shape[2]
        bid = C.c_int32(-1)
        self._check(self._lib.extrack_attach_bucket(self._h, C.c_void_p(tracks_t.data_ptr()), N, L, D,
                                                     C.c_void_p(sigma_t.data_ptr()) if sigma_t is not None else None, KS, C.byref(bid)))
        self._keep.append((tracks_t, sigma_t))
        self.buckets.append((N, L, D, KS))
        return bid.value

    def clear_buckets(self):
        self._check(self._lib.extrack_clear_buckets(self._h))
        self.buckets, self._keep = [], []

    def n_tracks(self):
        return sum(b[0] for b in self.buckets)

    def loglik(self, model, per_track=False):
        tot = C.c_double(0.0)
        out = np.empty(self.n_tracks()) if per_track else None
        self._check(self._lib.extrack_loglik(self._h, C.byref(model.c), C.byref(tot), out.ctypes.data_as(C.c_void_p) if per_track else None))
        return (tot.value, out) if per_track else tot.value

    @staticmethod
    def _pack_tangents(model, tangents):
        """(n, ctypes array of extrack_model_tangent, keep-alive list) from the packed dict of gradient.model_tangents (arrays with the
        direction as first axis: ds2 [n, S], Fs [n, S], TrMat [n, S, S], p_stay [n, G] and optionally locerr [n, <= 3], slope, offset,
        pBL [n]) or from a list of per-direction dicts with the same keys."""
        S = model.c.n_states
        G = model.p_stay.shape[-1]
        if isinstance(tangents, dict):
            n = len(np.atleast_1d(tangents["pBL"])) if "pBL" in tangents else len(tangents["ds2"])
            get = lambda k, shp: _f64(np.broadcast_to(np.asarray(tangents.get(k, 0.0), float), (n,) + shp))
        else:
            n = len(tangents)
            get = lambda k, shp: _f64(np.array([np.broadcast_to(np.asarray(t.get(k, 0.0), float), shp) for t in tangents]).reshape((n,) + shp))
        arr = (ExtrackModelTangent * max(n, 1))()
        keep = []
        if n:
            le = np.zeros((n, 3))
            if isinstance(tangents, dict):
                v = np.asarray(tangents.get("locerr", np.zeros((n, 0))), float).reshape(n, -1)
                le[:, :v.shape[1]] = v
            else:
                for i, t in enumerate(tangents):
                    v = np.atleast_1d(np.asarray(t.get("locerr", 0.0), float)).ravel()
                    le[i, :len(v)] = v
            sl, of, pb = get("slope", ()), get("offset", ()), get("pBL", ())
            keep = [get("ds2", (S,)), get("Fs", (S,)), get("TrMat", (S, S)), get("p_stay", (G,))]
            base = [x.ctypes.data for x in keep]
            step = [x.strides[0] for x in keep]
            for i in range(n):
                e = arr[i]
                e.locerr[0], e.locerr[1], e.locerr[2] = le[i]
                e.slope, e.offset, e.pBL = sl[i], of[i], pb[i]
                e.ds2, e.Fs, e.TrMat, e.p_stay = [C.cast(b + i * st, _dp) for b, st in zip(base, step)]
        return n, arr, keep

    def loglik_grad(self, model, tangents):
        """(sum LL, d sum LL / d theta_i) for the model directions ``tangents`` (see ``_pack_tangents``)."""
        n, arr, keep = self._pack_tangents(model, tangents)
        tot = C.c_double(0.0)
        g = np.zeros(max(n, 1))
        self._check(self._lib.extrack_loglik_grad(self._h, C.byref(model.c), n, arr, C.byref(tot), g.ctypes.data_as(C.c_void_p)))
        return tot.value, g[:n]

    def loglik_grad_async(self, model, tangents, d_out_ptr):
        """Enqueues the evaluation on the context's stream; the DEVICE buffer ``d_out_ptr`` (1 + n doubles) receives
        {sum LL, gradient} in stream order (the multi-GPU objective all-reduces it there)."""
        n, arr, keep = self._pack_tangents(model, tangents)
        self._check(self._lib.extrack_loglik_grad_async(self._h, C.byref(model.c), n, arr, C.c_void_p(d_out_ptr)))
        return n

    def sequence_matrix_th(self, model, bucket_id, threshold=0.2, max_nb_states=120):
        """LP[N, n_cols] of the threshold-fusion kernel for one bucket taken as one chunk, without the leaving term (extrack_sequence_matrix_th)."""
        N = self.buckets[bucket_id][0]
        nc = C.c_int64(0)
        self._check(self._lib.extrack_sequence_matrix_th(self._h, C.byref(model.c), int(bucket_id), C.c_double(threshold), int(max_nb_states), None, 0,
                                                         C.byref(nc)))
        out = np.empty((N, nc.value))
        self._check(self._lib.extrack_sequence_matrix_th(self._h, C.byref(model.c), int(bucket_id), C.c_double(threshold), int(max_nb_states),
                                                         out.ctypes.data_as(C.c_void_p), nc.value, C.byref(nc)))
        return out

    def th_freeze_plan(self, on):
        """Threshold-fusion evaluations follow the plan of the last planning evaluation (on) / decide their own again (off)."""
        self._check(self._lib.extrack_th_freeze_plan(self._h, 1 if on else 0))

    def loglik_th_grad(self, model, tangents, threshold=0.2, max_nb_states=120, chunk=2000):
        """(sum LL, d sum LL / d theta_i) of the threshold-fusion objective at the frozen plan of this evaluation (extrack_loglik_th_grad)."""
        n, arr, keep = self._pack_tangents(model, tangents)
        tot = C.c_double(0.0)
        g = np.zeros(max(n, 1))
        self._check(self._lib.extrack_loglik_th_grad(self._h, C.byref(model.c), C.c_double(threshold), int(max_nb_states), int(chunk), n, arr,
                                                     C.byref(tot), g.ctypes.data_as(C.c_void_p)))
        return tot.value, g[:n]

    def loglik_th_grad_async(self, model, tangents, threshold, max_nb_states, chunk, d_out_ptr):
        """Enqueues the evaluation; the DEVICE buffer ``d_out_ptr`` (1 + n doubles) receives {sum LL, gradient} in stream order."""
        n, arr, keep = self._pack_tangents(model, tangents)
        self._check(self._lib.extrack_loglik_th_grad_async(self._h, C.byref(model.c), C.c_double(threshold), int(max_nb_states), int(chunk), n, arr,
                                                           C.c_void_p(d_out_ptr)))
        return n

    def segment_len_hist(self, model, bucket_id, max_nb_states=500):
        """State-duration histogram [len - 1, S] of one bucket (extrack/histograms.py:26-286 semantics, see include/extrack_hip.h)."""
        N, L, D, KS = self.buckets[bucket_id]
        out = np.zeros((L - 1, model.c.n_states))
        self._check(self._lib.extrack_segment_len_hist(self._h, C.byref(model.c), int(bucket_id), int(max_nb_states), out.ctypes.data_as(C.c_void_p)))
        return out

    def refine_positions(self, model, bucket_id, threshold=0.1, max_nb_states=1000):
        """Refined positions [n, len, dims] and their stds [n, len] of one bucket (extrack/refined_localization.py:304-338 semantics)."""
        N, L, D, KS = self.buckets[bucket_id]
        mu, sg = np.zeros((N, L, D)), np.zeros((N, L))
        self._check(self._lib.extrack_refine_positions(self._h, C.byref(model.c), int(bucket_id), C.c_double(threshold), int(max_nb_states),
                                                       mu.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p)))
        return mu, sg

    def refine_pos_pdf(self, model, bucket_id, threshold=0.1, max_nb_states=1000):
        """The Gaussian mixture of every position of one bucket (get_pos_PDF, extrack/refined_localization.py:207-298): three lists over the
        positions of means [n, n_comp, dims], stds [n, n_comp, 1] and log-weights [n, n_comp]."""
        N, L, D, KS = self.buckets[bucket_id]
        counts = np.zeros(L, np.int32)
        args = (self._h, C.byref(model.c), int(bucket_id), C.c_double(threshold), int(max_nb_states), counts.ctypes.data_as(C.c_void_p))
        self._check(self._lib.extrack_refine_pos_pdf(*args, 0, None, None, None))  # sizing call
        tot = int(counts.sum())
        means, stds, logw = np.zeros((tot, N, D)), np.zeros((tot, N)), np.zeros((tot, N))
        self._check(self._lib.extrack_refine_pos_pdf(*args, tot, means.ctypes.data_as(C.c_void_p), stds.ctypes.data_as(C.c_void_p),
                                                     logw.ctypes.data_as(C.c_void_p)))
        off = np.concatenate(([0], np.cumsum(counts)))
        cut = lambda a, k: a[off[k]:off[k + 1]]
        return ([np.ascontiguousarray(cut(means, k).transpose(1, 0, 2)) for k in range(L)],
                [np.ascontiguousarray(cut(stds, k).T)[:, :, None] for k in range(L)],
                [np.ascontiguousarray(cut(logw, k).T) for k in range(L)])

    def last_grad_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.extrack_last_grad_ms(self._h, C.byref(ms)))
        return ms.value

    def loglik_th(self, model, threshold=0.2, max_nb_states=120, chunk=2000, per_track=False):
        """Threshold-fusion log-likelihood (extrack/tracking.py:427-743 semantics, see include/extrack_hip.h)."""
        tot = C.c_double(0.0)
        out = np.empty(self.n_tracks()) if per_track else None
        self._check(self._lib.extrack_loglik_th(self._h, C.byref(model.c), C.c_double(threshold), int(max_nb_states), int(chunk),
                                                C.byref(tot), out.ctypes.data_as(C.c_void_p) if per_track else None))
        return (tot.value, out) if per_track else tot.value

    def loglik_th_async(self, model, threshold, max_nb_states, chunk, d_total_ptr=None):
        self._check(self._lib.extrack_loglik_th_async(self._h, C.byref(model.c), C.c_double(threshold), int(max_nb_states), int(chunk),
                                                      C.c_void_p(d_total_ptr) if d_total_ptr else None))

    def predict_th(self, model, bucket_id, threshold=0.1, max_nb_states=200, nb_max=1):
        """Threshold-fusion posteriors of one bucket in chunks of nb_max tracks (extrack/tracking.py:792-906 semantics)."""
        N, L, D, KS = self.buckets[bucket_id]
        out = np.empty((N, L, model.c.n_states))
        self._check(self._lib.extrack_predict_th(self._h, C.byref(model.c), int(bucket_id), C.c_double(threshold), int(max_nb_states),
                                                 int(nb_max), out.ctypes.data_as(C.c_void_p)))
        return out

    def th_plan_step(self, bucket_id, chunk_index, t):
        """Merge groups (list of index arrays) decided at step t for one chunk by the last loglik_th call."""
        nE, nG = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.extrack_th_plan_step(self._h, int(bucket_id), int(chunk_index), int(t), C.byref(nE), C.byref(nG), None, None, 0))
        if nG.value == 0:
            return nE.value, []
        mem = np.zeros(nE.value, np.uint16)
        gst = np.zeros(nG.value + 1, np.uint16)
        self._check(self._lib.extrack_th_plan_step(self._h, int(bucket_id), int(chunk_index), int(t), C.byref(nE), C.byref(nG),
                                                   mem.ctypes.data_as(C.c_void_p), gst.ctypes.data_as(C.c_void_p), max(len(mem), len(gst))))
        return nE.value, [mem[gst[g]:gst[g + 1]].astype(int) for g in range(nG.value)]

    def loglik_async(self, model, d_total_ptr=None):
        self._check(self._lib.extrack_loglik_async(self._h, C.byref(model.c), C.c_void_p(d_total_ptr) if d_total_ptr else None))

    def predict(self, model, bucket_id):
        N, L, D, KS = self.buckets[bucket_id]
        out = np.empty((N, L, model.c.n_states))
        self._check(self._lib.extrack_predict(self._h, C.byref(model.c), int(bucket_id), out.ctypes.data_as(C.c_void_p)))
        return out

    def sequence_matrix(self, model, bucket_id):
        """LP[N, nB]: log-probability of every sequence of states still distinguished at the last position, in the reference's column
        order (first return value of P_Cs_inter_bound_stats, extrack/tracking.py:318)."""
        N, L, D, KS = self.buckets[bucket_id]
        m = model.c
        nb = int(self._lib.extrack_sequence_columns(m.n_states, L, m.nb_substeps, m.frame_len, int(L != m.max_len)))
        if nb < 1:
            raise ValueError("sequence matrix: invalid model / track length")
        out = np.empty((N, nb))
        self._check(self._lib.extrack_sequence_matrix(self._h, C.byref(m), int(bucket_id), out.ctypes.data_as(C.c_void_p), nb))
        return out

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.extrack_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_launch_info(self):
        info = (C.c_int32 * 6)()
        self._check(self._lib.extrack_last_launch_info(self._h, C.byref(info)))
        return dict(zip(("blocks", "threads", "lds_bytes", "tracks_per_block", "blocks_per_cu", "compute_units"), list(info)))


class MultiContext:
    """One process, several GPUs (extrack_multi_* of include/extrack_hip.h): every device keeps a contiguous row range of every bucket, an
    evaluation ends with one all-reduce of the scalar over RCCL (``use_rccl``: 0 host sum, 1 RCCL when loadable, 2 RCCL or error).  The
    multi-process form of the same partitioning is ``extrack_amd.distributed``."""

    def __init__(self, devices, use_rccl=1):
        self._lib = load()
        ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.extrack_multi_create(len(devices), ids, int(use_rccl), C.byref(h))
        if rc != 0:
            raise ExtrackError(rc, self._lib.extrack_multi_last_error(None).decode())
        self._h = h
        self.devices = [int(d) for d in devices]

    def _check(self, rc):
        if rc != 0:
            raise ExtrackError(rc, self._lib.extrack_multi_last_error(self._h).decode())

    @property
    def uses_rccl(self):
        return bool(self._lib.extrack_multi_uses_rccl(self._h))

    def upload_bucket(self, tracks, sigma=None):
        tracks = _f64(tracks)
        N, L, D = tracks.shape
        KS, sp = 0, None
        if sigma is not None:
            sigma = _f64(sigma)
            KS, sp = sigma.shape[2], sigma.ctypes.data_as(C.c_void_p)
        self._check(self._lib.extrack_multi_upload_bucket(self._h, tracks.ctypes.data_as(C.c_void_p), N, L, D, sp, KS))

    def loglik(self, model):
        tot = C.c_double(0.0)
        self._check(self._lib.extrack_multi_loglik(self._h, C.byref(model.c), C.byref(tot)))
        return tot.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.extrack_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def p_stay_table_c(ds, nb_substeps, cell_dims):
    """The library's own erfc-based p_stay table (host code, no GPU needed)."""
    lib = load()
    ds = _f64(ds)
    cd = _f64(cell_dims)
    out = np.empty(len(ds) ** int(nb_substeps))
    rc = lib.extrack_p_stay_table(ds.ctypes.data_as(C.c_void_p), len(ds), int(nb_substeps), cd.ctypes.data_as(C.c_void_p), len(cd),
                                  out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ExtrackError(rc, "extrack_p_stay_table")
    return out
