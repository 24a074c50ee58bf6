"""ctypes driver for tests/emul/libxt_emul.so (test infrastructure)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libxt_emul.so")
        src = os.path.join(HERE, "emul.cpp")
        hdrs = [os.path.join(HERE, "..", "..", "extrack_amd", "csrc", h) for h in ("xt_kernel.h", "xt_math.h", "xt_tables.h", "xt_dispatch.h", "xt_th.h", "xt_entry.h", "xt_fast2.h", "xt_grad.h",
                                                                                             "xt_grad_host.h", "xt_thgrad.h", "xt_thgrad2.h", "xt_big.h", "xt_hist.h", "xt_hist_host.h", "xt_reg2.h", "xt_gradr.h", "xt_rev.h", "xt_seqmat.h")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src, os.path.join(HERE, "emul_r2.cpp"), os.path.join(HERE, "emul_gradr.cpp"), os.path.join(HERE, "emul_rev.cpp"), os.path.join(HERE, "emul_ctx.h")] + hdrs):
            import subprocess
            units = ["emul.cpp", "emul_r2.cpp", "emul_gradr.cpp", "emul_rev.cpp"]  # compiled side by side: emul_r2.cpp unrolls the whole step loop per instance
            procs = [subprocess.Popen(["g++", "-O1", "-std=c++17", "-fPIC", "-pthread", "-c", os.path.join(HERE, u), "-o",
                                       os.path.join(HERE, u[:-4] + ".o")]) for u in units]
            if any(p.wait() != 0 for p in procs):
                raise RuntimeError("g++ failed on tests/emul")
            subprocess.check_call(["g++", "-shared", "-pthread", "-o", so] + [os.path.join(HERE, u[:-4] + ".o") for u in units])
        _lib = C.CDLL(so)
    return _lib


def dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def run(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len, preds=False, nblocks=2, slope=None, offset=None):
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    LE = np.ascontiguousarray(LE, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS = 0, LE.shape[2], 1
        locerr = np.zeros(3)
        locerr[:K] = LE[0, 0]
        sigma = None
    else:
        mode, KS = (2 if slope is not None else 1), LE.shape[2]
        K, locerr, sigma = KS, np.zeros(3), LE
    ll = np.zeros(N)
    pr = np.full((N, L, S), -1.0) if preds else None
    tot = C.c_double(0)
    info = (C.c_int * 4)()
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_run(dp(Cs), dp(sigma), C.c_longlong(N), L, D, KS, S, ns, F, int(isBL), int(min_len), mode, K, dp(locerr),
                           C.c_double(slope or 0.0), C.c_double(offset or 0.0), C.c_double(pBL), dp(ds), dp(Fs), dp(T), dp(p_stay),
                           int(preds), nblocks, dp(ll), dp(pr), C.byref(tot), info)
    if rc != 0:
        raise RuntimeError("emul rc=%d" % rc)
    return ll, pr, tot.value, list(info)


def run_seq_matrix(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len):
    """Per-sequence log-probabilities LP[N, nB] in the reference's column order (P_Cs_inter_bound_stats' first return value): the general
    kernel body with the raw per-(sequence, new digits) output, then the host mapping of csrc/xt_seqmat.h."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    G, E = S ** ns, S ** F
    raw = np.full((N, E, G), np.nan)
    lib().xt_emul_set_seq_out(dp(raw))
    run(Cs, LE, ds, Fs, T, pBL, 0, p_stay, ns, F, min_len)  # without the leaving term: the mapping adds it per expanded sequence
    lib().xt_emul_seq_columns.restype = C.c_longlong
    nb = lib().xt_emul_seq_columns(S, L, ns, F, int(isBL))
    lp = np.zeros((N, nb))
    T, p_stay = np.ascontiguousarray(T, float), np.ascontiguousarray(p_stay, float)
    rc = lib().xt_emul_seq_reorder(S, ns, F, C.c_longlong(N), L, int(isBL), C.c_double(pBL), dp(T), dp(p_stay), dp(raw), dp(lp))
    if rc != 0:
        raise RuntimeError("emul seq reorder rc=%d" % rc)
    return lp


def run_multi(buckets, locerr, ds, Fs, T, pBL, p_stay, ns, F, min_len, max_len, blocks_per_bucket):
    """buckets: list of arrays [N, L, D]; one emulated launch over all of them.  Returns (per-track lists, total)."""
    nb = len(buckets)
    bk = [np.ascontiguousarray(b, float) for b in buckets]
    D = bk[0].shape[2]
    S = len(ds)
    le = np.zeros(3)
    locerr = np.atleast_1d(np.asarray(locerr, float)).ravel()
    le[:len(locerr)] = locerr
    outs = [np.zeros(len(b)) for b in bk]
    PD = C.POINTER(C.c_double)
    tr = (PD * nb)(*[dp(b) for b in bk])
    lo = (PD * nb)(*[dp(o) for o in outs])
    Ns = (C.c_longlong * nb)(*[len(b) for b in bk])
    Ls = (C.c_int * nb)(*[b.shape[1] for b in bk])
    bp = (C.c_int * nb)(*blocks_per_bucket)
    tot = C.c_double(0)
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_run_multi(nb, tr, Ns, Ls, D, S, ns, F, int(max_len), int(min_len), len(locerr), dp(le), C.c_double(pBL), dp(ds),
                                 dp(Fs), dp(T), dp(p_stay), bp, lo, C.byref(tot))
    if rc != 0:
        raise RuntimeError("emul multi rc=%d" % rc)
    return outs, tot.value


def run_th(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len, threshold, max_nb, chunk=2000, capE=256, TT=8, threads=64, nblocks=2,
           slope=None, offset=None):
    """Threshold-fusion plan + apply kernel bodies on CPU threads.  Returns (per-track LL, total, plan) with
    plan[chunk][t] = list of member arrays (groups) for the fused steps."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    LE = np.ascontiguousarray(LE, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS = 0, LE.shape[2], 0
        locerr = np.zeros(3)
        locerr[:K] = LE[0, 0]
        sigma = None
    else:
        mode, KS = (2 if slope is not None else 1), LE.shape[2]
        K, locerr, sigma = KS, np.zeros(3), LE
    nch = (N + chunk - 1) // chunk
    ll = np.zeros(N)
    tot = C.c_double(0)
    hdr = np.zeros((nch, L, 2), np.int32)
    mem = np.zeros((nch, L, capE), np.uint16)
    gst = np.zeros((nch, L, capE + 1), np.uint16)
    status = np.zeros((nch, 4), np.int32)
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    vp = C.c_void_p
    rc = lib().xt_emul_th_run(dp(Cs), dp(sigma), C.c_longlong(N), L, D, KS, S, ns, F, int(isBL), int(min_len), mode, K, dp(locerr),
                              C.c_double(slope or 0.0), C.c_double(offset or 0.0), C.c_double(pBL), dp(ds), dp(Fs), dp(T), dp(p_stay),
                              C.c_double(threshold), int(max_nb), int(chunk), int(capE), int(TT), int(threads), int(nblocks), dp(ll),
                              C.byref(tot), hdr.ctypes.data_as(vp), mem.ctypes.data_as(vp), gst.ctypes.data_as(vp), status.ctypes.data_as(vp))
    if rc != 0:
        raise RuntimeError("emul th rc=%d status=%s" % (rc, status.tolist()))
    plan = []
    for c in range(nch):
        steps = {}
        for t in range(2, L - 1):
            nE, nG = hdr[c, t]
            steps[t] = [mem[c, t, gst[c, t, g]:gst[c, t, g + 1]].astype(int) for g in range(nG)]
        plan.append(steps)
    return ll, tot.value, plan, hdr, status


def pack_tangents(tangents, S, G):
    """Rows [locerr(3), slope, offset, pBL, ds2(S), Fs(S), TrMat(S*S), p_stay(G)] of a list of tangent dicts."""
    rows = []
    for t in tangents:
        le = np.zeros(3)
        v = np.atleast_1d(np.asarray(t.get("locerr", 0.0), float)).ravel()
        le[:len(v)] = v
        rows.append(np.concatenate([le, [t.get("slope", 0.0), t.get("offset", 0.0), t.get("pBL", 0.0)],
                                    np.broadcast_to(np.asarray(t.get("ds2", 0.0), float), (S,)),
                                    np.broadcast_to(np.asarray(t.get("Fs", 0.0), float), (S,)),
                                    np.broadcast_to(np.asarray(t.get("TrMat", 0.0), float), (S, S)).ravel(),
                                    np.broadcast_to(np.asarray(t.get("p_stay", 0.0), float), (G,))]))
    return np.ascontiguousarray(np.array(rows, float)) if rows else np.zeros((1, 1))


def run_th_grad(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len, threshold, max_nb, tangents, waves=1, **kw):
    """Threshold-fusion plan + apply bodies, then the frozen-plan gradient body (xt_thgrad.h) on the plan just made.
    Returns (apply LL per track, gradient-body LL per track, gradient-body total, gradient[n_dir], plan)."""
    N = len(Cs)
    S = len(ds)
    tan = pack_tangents(tangents, S, S ** ns)
    out = np.zeros(len(tangents) + 1)
    llg = np.zeros(N)
    lib().xt_emul_th_set_grad(len(tangents), dp(tan), dp(out), dp(llg), int(waves))
    ll, tot, plan, hdr, status = run_th(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len, threshold, max_nb, **kw)
    return ll, llg, out[0], out[1:], plan


def run_th_predict(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, F, min_len, threshold, max_nb, chunk=1, capE=256, threads=64, nblocks=2,
                   slope=None, offset=None):
    """Threshold-fusion posteriors (plan body in prediction mode) on CPU threads: preds[N, L, S]."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    LE = np.ascontiguousarray(LE, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS = 0, LE.shape[2], 0
        locerr = np.zeros(3)
        locerr[:K] = LE[0, 0]
        sigma = None
    else:
        mode, KS = (2 if slope is not None else 1), LE.shape[2]
        K, locerr, sigma = KS, np.zeros(3), LE
    nch = (N + chunk - 1) // chunk
    pr = np.full((N, L, S), np.nan)
    status = np.zeros((nch, 4), np.int32)
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_th_predict(dp(Cs), dp(sigma), C.c_longlong(N), L, D, KS, S, F, int(isBL), int(min_len), mode, K, dp(locerr),
                                  C.c_double(slope or 0.0), C.c_double(offset or 0.0), C.c_double(pBL), dp(ds), dp(Fs), dp(T), dp(p_stay),
                                  C.c_double(threshold), int(max_nb), int(chunk), int(capE), int(threads), int(nblocks), dp(pr),
                                  status.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("emul th predict rc=%d status=%s" % (rc, status.tolist()))
    return pr


def run_th_multi(buckets, locerr, ds, Fs, T, pBL, p_stay, ns, F, min_len, max_len, threshold, max_nb, chunk, capE=128, TT=8, threads=128, bpc=2):
    """Several buckets through the bucket-descriptor table: one emulated plan launch + one apply launch."""
    nb = len(buckets)
    bk = [np.ascontiguousarray(b, float) for b in buckets]
    D = bk[0].shape[2]
    S = len(ds)
    le = np.zeros(3)
    locerr = np.atleast_1d(np.asarray(locerr, float)).ravel()
    le[:len(locerr)] = locerr
    outs = [np.zeros(len(b)) for b in bk]
    PD = C.POINTER(C.c_double)
    tr = (PD * nb)(*[dp(b) for b in bk])
    lo = (PD * nb)(*[dp(o) for o in outs])
    Ns = (C.c_longlong * nb)(*[len(b) for b in bk])
    Ls = (C.c_int * nb)(*[b.shape[1] for b in bk])
    tot = C.c_double(0)
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_th_run_multi(nb, tr, Ns, Ls, D, S, ns, F, int(max_len), int(min_len), len(locerr), dp(le), C.c_double(pBL), dp(ds),
                                    dp(Fs), dp(T), dp(p_stay), C.c_double(threshold), int(max_nb), int(chunk), int(capE), int(TT),
                                    int(threads), int(bpc), lo, C.byref(tot))
    if rc != 0:
        raise RuntimeError("emul th multi rc=%d" % rc)
    return outs, tot.value


def run_grad(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, ns, F, min_len, tangents, nblocks=2, tpb=2, tan_lds=1, generic_g=0, PJ=1, slope=None,
             offset=None):
    """Likelihood + gradient body on CPU threads.  tangents: list of dicts (keys ds2, Fs, TrMat, p_stay, locerr, slope, offset, pBL).
    Returns (per-track LL, total LL, gradient[n_dir])."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    G = S ** ns
    LE = np.ascontiguousarray(LE, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS = 0, LE.shape[2], 1
        locerr = np.zeros(3)
        locerr[:K] = LE[0, 0]
        sigma = None
    else:
        mode, KS = (2 if slope is not None else 1), LE.shape[2]
        K, locerr, sigma = KS, np.zeros(3), LE
    rows = []
    for t in tangents:
        le = np.zeros(3)
        v = np.atleast_1d(np.asarray(t.get("locerr", 0.0), float)).ravel()
        le[:len(v)] = v
        rows.append(np.concatenate([le, [t.get("slope", 0.0), t.get("offset", 0.0), t.get("pBL", 0.0)],
                                    np.broadcast_to(np.asarray(t.get("ds2", 0.0), float), (S,)),
                                    np.broadcast_to(np.asarray(t.get("Fs", 0.0), float), (S,)),
                                    np.broadcast_to(np.asarray(t.get("TrMat", 0.0), float), (S, S)).ravel(),
                                    np.broadcast_to(np.asarray(t.get("p_stay", 0.0), float), (G,))]))
    tan = np.ascontiguousarray(np.array(rows, float)) if rows else np.zeros((1, 1))
    ll = np.zeros(N)
    out = np.zeros(len(tangents) + 1)
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_grad(dp(Cs), dp(sigma), C.c_longlong(N), L, D, KS, S, ns, F, int(isBL), int(min_len), mode, K, dp(locerr),
                            C.c_double(slope or 0.0), C.c_double(offset or 0.0), C.c_double(pBL), dp(ds), dp(Fs), dp(T), dp(p_stay),
                            len(tangents), dp(tan), nblocks, tpb, tan_lds, generic_g, PJ, dp(ll), dp(out))
    if rc != 0:
        raise RuntimeError("emul grad rc=%d" % rc)
    return ll, out[0], out[1:]


def run_hist(Cs, LE, ds, Fs, T, pBL, isBL, p_stay, min_l, max_nb_states, nblocks=2, threads=64, par_lds=1, slope=None, offset=None):
    """State-duration histogram body on CPU threads: returns hist[L - 1, S] summed over the tracks."""
    Cs = np.ascontiguousarray(Cs, float)
    N, L, D = Cs.shape
    S = len(ds)
    LE = np.ascontiguousarray(LE, float)
    if LE.shape[1] == 1 and L != 1:
        mode, K, KS = 0, LE.shape[2], 1
        locerr = np.zeros(3)
        locerr[:K] = LE[0, 0]
        sigma = None
    else:
        mode, KS = (2 if slope is not None else 1), LE.shape[2]
        K, locerr, sigma = KS, np.zeros(3), np.ascontiguousarray(np.broadcast_to(LE, (N, L, LE.shape[2])))
    out = np.zeros((L - 1, S))
    ds, Fs, T, p_stay = [np.ascontiguousarray(x, float) for x in (ds, Fs, T, p_stay)]
    rc = lib().xt_emul_hist(dp(Cs), dp(sigma), C.c_longlong(N), L, D, KS, S, int(isBL), int(min_l), mode, K, dp(locerr), C.c_double(slope or 0.0),
                            C.c_double(offset or 0.0), C.c_double(pBL), dp(ds), dp(Fs), dp(T), dp(p_stay), int(max_nb_states), nblocks, threads,
                            par_lds, dp(out))
    if rc != 0:
        raise RuntimeError("emul hist rc=%d" % rc)
    return out


def set_th_dt(dt, p_stay_chunks):
    """Per-track time steps for the NEXT run_th / run_th_predict call: dt [N, L], one p_stay table [G] per chunk (keep both alive)."""
    lib().xt_emul_th_set_dt(dp(dt), dp(p_stay_chunks))
