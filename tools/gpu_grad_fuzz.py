"""Differential fuzz of the gradient kernel families on the GPU box: the launcher's choice (xt_reg2.h / xt_rev.h / xt_gradr.h) - or the family
named by the third argument (EXTRACK_GRAD_PATH value: rev, gradr, reg2) - against the LDS-resident kernel (xt_grad.h, itself pinned to Richardson
differences of the oracle in tests/test_hip_grad.py) on random models, random DENSE tangent directions, all localisation-error modes, lengths
around the window.  usage: gpu_grad_fuzz.py [seed] [cases] [path]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import tracking as T  # noqa: E402
from oracle import oracle_np as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
other = sys.argv[3] if len(sys.argv) > 3 else None
rng = np.random.default_rng(seed)
t0 = time.time()
bad = done = 0
worst = worst_ll = 0.0
for case in range(ncases):
    S = int(rng.choice([2, 2, 2, 3, 3, 4]))
    ns = int(rng.choice([1, 1, 1, 2]))
    F = int(rng.integers(ns + 1, 9))
    if S ** F > 1024 or S ** ns > 9:
        continue
    D = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.integers(2, 40))
    N = int(rng.choice([1, 3, 17, 64, 130]))
    kind = str(rng.choice(["scalar", "scalar", "perdim", "peak1", "peakD", "affine"]))
    if D == 1 and kind in ("perdim", "peakD"):
        kind = "scalar"
    min_len = int(rng.choice([2, 3, 5]))
    max_len = L + int(rng.integers(0, 2))  # isBL 0 / 1
    pBL = float(rng.uniform(0.01, 0.3))
    cell = [float(rng.uniform(0.3, 2.0))]
    ds = np.sort(rng.uniform(0.004, 0.25, S))
    Fs = rng.dirichlet(np.ones(S) * 2)
    Tm = rng.uniform(0.01, 0.9 / S, (S, S))
    Tm[np.arange(S), np.arange(S)] = 0
    Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
    Cs = np.cumsum(rng.normal(0, 1, (N, L, D)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, D))
    sig, so, LE = None, None, None
    if kind == "scalar":
        LE = np.array([[[0.02]]])
    elif kind == "perdim":
        LE = rng.uniform(0.012, 0.03, (1, 1, D))
    else:
        sig = rng.uniform(0.012, 0.035, (N, L, 1 if kind == "peak1" else D))
        if kind == "affine":
            so = (float(rng.uniform(0.8, 1.3)), float(rng.uniform(-0.002, 0.004)))
    ndir = int(rng.integers(1, 15))
    G = S ** ns
    tang = dict(ds2=rng.normal(0, 1e-3, (ndir, S)), Fs=rng.normal(0, 0.1, (ndir, S)), TrMat=rng.normal(0, 0.02, (ndir, S, S)),
                pBL=rng.normal(0, 0.05, ndir), p_stay=rng.normal(0, 0.01, (ndir, G)))
    if rng.random() < 0.3:  # some "uniform" directions (only pBL moves)
        k = int(rng.integers(0, ndir))
        for key in ("ds2", "Fs", "TrMat", "p_stay"):
            tang[key][k] = 0.0
    if sig is None:
        tang["locerr"] = rng.normal(0, 1e-3, (ndir, LE.shape[2]))
    elif so is not None:
        tang["slope"], tang["offset"] = rng.normal(0, 0.1, ndir), rng.normal(0, 1e-3, ndir)
    cfg = dict(S=S, ns=ns, F=F, D=D, L=L, N=N, kind=kind, isBL=int(L != max_len), min_len=min_len, ndir=ndir)
    out = {}
    try:
        for path in ("lds", "default"):
            os.environ.pop("EXTRACK_GRAD_PATH", None)
            if path == "lds":
                os.environ["EXTRACK_GRAD_PATH"] = "lds"
            elif other:
                os.environ["EXTRACK_GRAD_PATH"] = other
            ts = T.TrackSet([Cs], None if sig is None else [sig], min_len=min(min_len, L), max_len=max_len)
            model = ts.make_model(LE, ds, Fs, Tm, pBL, cell, ns, F, slope_offset=so)
            try:
                out[path] = ts.ctx.loglik_grad(model, tang)
            except Exception as e:  # noqa: BLE001
                if path == "lds" and "does not fit" in str(e):  # a capacity limit of the reference kernel family: compare with central differences instead
                    out[path] = None
                else:
                    raise
            finally:
                ts.close()
        if out["lds"] is None:
            def total(x):
                return O.proba_cs(Cs, (LE + x * tang["locerr"][0][None, None]) if sig is None else (sig if so is None else np.clip(sig * (so[0] + x * tang["slope"][0]) + so[1] + x * tang["offset"][0], 1e-6, np.inf)),
                                  np.sqrt(ds ** 2 + x * tang["ds2"][0]), Fs + x * tang["Fs"][0], Tm + x * tang["TrMat"][0], pBL + x * tang["pBL"][0], int(L != max_len), cell, ns, F, min(min_len, L)).sum()
            # p_stay moves with its own tangent in the kernel; the oracle recomputes it from ds: use a direction without a p_stay tangent
            tang0 = {k: np.array(v[:1]) for k, v in tang.items()}
            tang0["p_stay"][:] = 0.0
            tang0["ds2"][:] = 0.0
            tang["ds2"][0] = 0.0
            ts = T.TrackSet([Cs], None if sig is None else [sig], min_len=min(min_len, L), max_len=max_len)
            ll1, g1 = ts.ctx.loglik_grad(ts.make_model(LE, ds, Fs, Tm, pBL, cell, ns, F, slope_offset=so), tang0)
            ts.close()
            h = 1e-6
            fd = (8 * (total(h / 2) - total(-h / 2)) - (total(h) - total(-h))) / (6 * h)
            out["lds"] = (ll1, np.array([fd]))
            out["default"] = (ll1, g1)
            tol_g = 1e-6
        else:
            tol_g = 1e-9
        ll0, g0 = out["lds"]
        ll1, g1 = out["default"]
        ref = O.proba_cs(Cs, LE if sig is None else (sig if so is None else np.clip(sig * so[0] + so[1], 1e-6, np.inf)), ds, Fs, Tm, pBL, int(L != max_len), cell, ns, F,
                         min(min_len, L)).sum()
        dl = abs(ll1 - ref) / max(1.0, abs(ref))
        dg = np.abs(g1 - g0).max() / max(np.abs(g0).max(), 1e-30)
        worst, worst_ll = max(worst, dg), max(worst_ll, dl)
        done += 1
        if not (dl < 1e-11 and dg < tol_g):
            bad += 1
            print("MISMATCH ll %.3e grad %.3e" % (dl, dg), cfg, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("EXCEPTION", repr(e)[:300], cfg, flush=True)
    if case % 50 == 0:
        print("case %d  done %d  bad %d  worst |dLL| %.2e  worst |dgrad| %.2e  (%.0f s)" % (case, done, bad, worst_ll, worst, time.time() - t0), flush=True)
print("DONE cases %d done %d bad %d worst rel LL %.3e worst rel grad %.3e in %.0f s" % (ncases, done, bad, worst_ll, worst, time.time() - t0))
