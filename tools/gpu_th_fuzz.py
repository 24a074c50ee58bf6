"""Differential fuzz of the threshold-fusion kernels against the pinned oracle on the GPU box: random models, track counts around
the pilot limit, chunk sizes, localisation-error modes, thresholds / max_nb_states; log-likelihoods and posteriors."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd.engine import TrackSet  # noqa: E402
from oracle import oracle_th as OT  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
t_start = time.time()
bad = 0
worst_ll = worst_pr = 0.0
for case in range(ncases):
    S = int(rng.choice([2, 2, 3, 3, 4, 5]))
    ns = int(rng.choice([1, 1, 1, 2, 3])) if S <= 3 else 1
    F = int(rng.integers(ns + 1, 8 if S == 2 else (6 if S == 3 else 5)))
    D = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.integers(2, 24))
    N = int(rng.choice([1, 2, 7, 29, 30, 31, 45, 64, 65, 100]))
    chunk = int(rng.choice([N, max(1, N // 2), 16, 31, 40]))
    kind = str(rng.choice(["scalar", "scalar", "perdim", "peak1", "peakD", "affine"]))
    if D == 1 and kind in ("perdim", "peakD"):
        kind = "scalar"
    thr = float(rng.choice([0.02, 0.1, 0.2, 0.5, 1.0]))
    mx = int(rng.choice([6, 30, 120, 1000]))
    isBL = int(rng.integers(0, 2))
    min_len = int(rng.choice([2, 3, 5]))
    pBL = float(rng.uniform(0.01, 0.3))
    cell = (float(rng.uniform(0.3, 2.0)),)
    ds = np.sort(rng.uniform(0.004, 0.25, S))
    Fs = rng.dirichlet(np.ones(S) * 2)
    Tm = rng.uniform(0.01, 0.9 / S, (S, S))
    Tm[np.arange(S), np.arange(S)] = 0
    Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
    Cs = np.cumsum(rng.normal(0, 1, (N, L, D)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, D))
    sig, so, le = None, None, np.array([[[0.02]]])
    if kind == "perdim":
        le = rng.uniform(0.012, 0.03, (1, 1, D))
    elif kind != "scalar":
        sig = rng.uniform(0.012, 0.035, (N, L, 1 if kind == "peak1" else D))
        so = (1.2, -0.003) if kind == "affine" else None
    sg = sig if so is None or sig is None else np.clip(sig * so[0] + so[1], 1e-6, np.inf)
    # skip cases whose live-sequence count would explode (oracle time)
    if S ** (ns + 1) * S ** ns > 3000:
        continue
    cfg = dict(S=S, ns=ns, F=F, D=D, L=L, N=N, chunk=chunk, kind=kind, thr=thr, mx=mx, isBL=isBL, min_len=min_len)
    try:
        ts = TrackSet([Cs], None if sig is None else [sig], min_len=min_len, max_len=L + 1 if isBL else L)
        try:
            model = ts.make_model(None if sig is not None else le, ds, Fs, Tm, pBL, cell, ns, F, slope_offset=so)
            _, ll = ts.loglik_th(model, thr, mx, chunk=chunk, per_track=True)
            ref = np.concatenate([OT.proba_cs_th(Cs[a:a + chunk], le if sg is None else sg[a:a + chunk], ds, Fs, Tm, pBL, isBL, cell, ns, F,
                                                 min_len, thr, mx) for a in range(0, N, chunk)])
            d = np.abs(ll - ref).max()
            # second evaluation on the same context: learned capacities (LDS-resident pilot state, larger workgroups, ...)
            _, ll2 = ts.loglik_th(model, thr, mx, chunk=chunk, per_track=True)
            d = max(d, np.abs(ll2 - ref).max())
            worst_ll = max(worst_ll, d)
            if not d < 1e-10:
                bad += 1
                print("LL MISMATCH", d, cfg, flush=True)
            if ns == 1 and N <= 64:
                nbm = int(rng.choice([1, 1, 3, min(30, N)]))
                m1 = ts.make_model(None if sig is not None else le, ds, Fs, Tm, pBL, cell, 1, F, slope_offset=so)
                pr = ts.predict_th(m1, thr, mx, nb_max=nbm)[0]
                prr = np.concatenate([OT.p_cs_inter_bound_stats_th(Cs[a:a + nbm], le if sg is None else sg[a:a + nbm], ds, Fs, Tm, pBL, isBL, cell, 1,
                                                                   F, 1, min_len, thr, mx)[1] for a in range(0, N, nbm)])
                dp = np.abs(pr - prr).max()
                worst_pr = max(worst_pr, dp)
                if not dp < 1e-9:
                    bad += 1
                    print("PRED MISMATCH", dp, "nb_max", nbm, cfg, flush=True)
        finally:
            ts.close()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("EXCEPTION", repr(e)[:300], cfg, flush=True)
    if case % 25 == 0:
        print("case %d  bad %d  worst LL %.2e  worst pred %.2e  (%.0f s)" % (case, bad, worst_ll, worst_pr, time.time() - t_start), flush=True)
print("DONE cases %d bad %d worst LL %.3e worst pred %.3e in %.0f s" % (ncases, bad, worst_ll, worst_pr, time.time() - t_start))
