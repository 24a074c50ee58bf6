"""ctypes driver of oracle/extrack_oracle.c (ORACLE = test infrastructure; see the C file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "extrack_oracle.c")
SO = os.path.join(HERE, "_build", "libextrack_oracle.so")
_lib = None


def build(force=False):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if force or not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", SO, "-lm"])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.extrack_oracle_run.restype = C.c_int
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def run(Cs, LocErr, ds, Fs, TrMat, pBL, isBL, p_stay, nb_substeps, frame_len, min_len, do_preds=False, slope=None, offset=None, nthreads=1):
    """Returns (LP_C[N], preds[N, L, S] or None).  LocErr: (1,1,k) global or (N,L,k) per peak, like the reference."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    ds, Fs, TrMat, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, TrMat, p_stay)]
    S = len(ds)
    LE = np.ascontiguousarray(LocErr, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS, sig = 0, LE.shape[2], 1, None
        loc = np.zeros(3)
        loc[:K] = LE[0, 0]
    else:
        mode, KS, sig, loc = (2 if slope is not None else 1), LE.shape[2], np.ascontiguousarray(np.broadcast_to(LE, (N, L, LE.shape[2]))), np.zeros(3)
        K = KS
    ll = np.zeros(N)
    pr = np.zeros((N, L, S)) if do_preds else None
    rc = lib().extrack_oracle_run(_dp(Cs), _dp(sig), C.c_long(N), L, D, KS, S, int(nb_substeps), int(frame_len), int(isBL), int(min_len), mode, K,
                                  _dp(loc), C.c_double(slope or 0.0), C.c_double(offset or 0.0), C.c_double(pBL), _dp(ds), _dp(Fs), _dp(TrMat),
                                  _dp(p_stay), _dp(ll), _dp(pr), int(nthreads))
    if rc != 0:
        raise RuntimeError("extrack_oracle_run rc=%d" % rc)
    return ll, pr
