"""Differential fuzz of the fixed-window kernels (fast 2-state path, general path, entry-parallel path, posteriors) against the
pinned numpy oracle on the GPU box: random models, dimensions, localisation-error modes, lengths around the window."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from extrack_amd import tracking as T  # noqa: E402
from oracle import oracle_np as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
t0 = time.time()
bad = 0
worst_ll = worst_pr = 0.0
for case in range(ncases):
    S = int(rng.choice([2, 2, 2, 3, 3, 4, 5]))
    ns = int(rng.choice([1, 1, 1, 2, 3]))
    F = int(rng.integers(ns + 1, 10))
    if S ** (F + ns) > 6000 or S ** F > 4096:
        continue
    D = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.integers(2, 40))
    N = int(rng.choice([1, 3, 17, 64, 130]))
    kind = str(rng.choice(["scalar", "scalar", "perdim", "peak1", "peakD"]))
    if D == 1 and kind in ("perdim", "peakD"):
        kind = "scalar"
    isBL = int(rng.integers(0, 2))
    min_len = int(rng.choice([2, 3, 5]))
    pBL = float(rng.uniform(0.01, 0.3))
    cell = [float(rng.uniform(0.3, 2.0))]
    ds = np.sort(rng.uniform(0.004, 0.25, S))
    Fs = rng.dirichlet(np.ones(S) * 2)
    Tm = rng.uniform(0.01, 0.9 / S, (S, S))
    Tm[np.arange(S), np.arange(S)] = 0
    Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
    Cs = np.cumsum(rng.normal(0, 1, (N, L, D)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, D))
    if kind == "scalar":
        LE = np.array([[[0.02]]])
    elif kind == "perdim":
        LE = rng.uniform(0.012, 0.03, (1, 1, D))
    else:
        LE = rng.uniform(0.012, 0.035, (N, L, 1 if kind == "peak1" else D))
    cfg = dict(S=S, ns=ns, F=F, D=D, L=L, N=N, kind=kind, isBL=isBL, min_len=min_len)
    try:
        got = T.Proba_Cs(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, ns, F, min_len)
        ref = O.proba_cs(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, ns, F, min_len)
        d = np.abs(got - ref).max()
        worst_ll = max(worst_ll, d)
        if not d < 1e-10:
            bad += 1
            print("LL MISMATCH", d, cfg, flush=True)
        if ns == 1 and S <= 5:
            _, _, pr = T.P_Cs_inter_bound_stats(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, 1, F, 1, min_len)
            _, prr = O.p_cs_inter_bound_stats(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, 1, F, 1, min_len)
            dp = np.abs(pr - prr).max()
            worst_pr = max(worst_pr, dp)
            if not dp < 1e-9:
                bad += 1
                print("PRED MISMATCH", dp, cfg, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("EXCEPTION", repr(e)[:300], cfg, flush=True)
    if case % 50 == 0:
        print("case %d  bad %d  worst LL %.2e  worst pred %.2e  (%.0f s)" % (case, bad, worst_ll, worst_pr, time.time() - t0), flush=True)
print("DONE cases %d bad %d worst LL %.3e worst pred %.3e in %.0f s" % (ncases, bad, worst_ll, worst_pr, time.time() - t0))
