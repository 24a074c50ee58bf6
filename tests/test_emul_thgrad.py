"""CPU checks of the frozen-plan gradient of the threshold-fusion objective (round 4; what extrack.tracking.param_fitting minimises in
v1.6.3: /root/reference/extrack/tracking.py:1371 -> :991 -> :427-743): the kernel body of extrack_amd/csrc/xt_thgrad.h run on CPU threads
(tests/emul) on the plan the plan body has just made, against Richardson-extrapolated central differences of the pinned numpy oracle
evaluated WITH THAT PLAN FROZEN (oracle_th ``plan=``) along every model direction."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))
from test_grad_cpu import _model, _richardson, model_directions  # noqa: E402


def _oracle_plan_and_fd(Cs, le_arr, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, thr, max_nb, dirs, chunk):
    from oracle import oracle_th as OT
    plans, base = [], []
    for a0 in range(0, len(Cs), chunk):
        tr = []
        base.append(OT.proba_cs_th(Cs[a0:a0 + chunk], le_arr(0.0, {})[a0:a0 + chunk] if le_arr(0.0, {}).shape[0] > 1 else le_arr(0.0, {}),
                                   np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len, thr, max_nb, trace=tr))
        plans.append(tr)

    def total(x, d):
        s = 0.0
        for ci, a0 in enumerate(range(0, len(Cs), chunk)):
            le = le_arr(x, d)
            le = le[a0:a0 + chunk] if le.shape[0] > 1 else le
            s += OT.proba_cs_th(Cs[a0:a0 + chunk], le, np.sqrt(ds2 + x * d.get("ds2", 0.0)), Fs + x * d.get("Fs", 0.0), T + x * d.get("T", 0.0),
                                pBL + x * d.get("pBL", 0.0), isBL, cell, ns, F, min_len, thr, max_nb, plan=plans[ci]).sum()
        return s

    fd = np.array([_richardson(lambda x: total(x, d), h) for _, _, d, h in dirs])
    return np.concatenate(base), plans, fd


@pytest.mark.parametrize("S,ns,F,L,N,D,K,isBL,chunk,waves", [(2, 1, 4, 9, 40, 2, 1, 1, 40, 1), (2, 1, 6, 14, 70, 2, 1, 0, 35, 2), (3, 1, 4, 8, 33, 2, 2, 1, 33, 1),
                                                               (2, 2, 3, 7, 20, 1, 1, 1, 20, 1), (3, 1, 5, 10, 12, 3, 3, 1, 12, 1), (2, 1, 5, 2, 9, 2, 1, 1, 9, 1),
                                                               (3, 1, 4, 3, 10, 2, 1, 0, 10, 1), (4, 1, 3, 7, 8, 3, 1, 1, 8, 1), (2, 1, 6, 40, 6, 2, 1, 1, 6, 1)])
def test_emulated_frozen_plan_gradient_vs_oracle_differences(S, ns, F, L, N, D, K, isBL, chunk, waves, monkeypatch):
    """Both mappings of the gradient body: one lane per track (xt_thgrad.h) and one lane per (sequence, track) with tiles of 4 / 16 tracks
    (xt_thgrad2.h: fewer lanes per track than sequences -> several passes; more table entries than lanes per track -> several per lane)."""
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F + L, dims=D)
    ds2, cell, pBL, min_len, thr, max_nb = 2 * Ds * 0.02, [1.0], 0.1, 3, 0.2, 120
    le = np.array([0.02, 0.025, 0.03][:K])
    dirs = model_directions(S, K, ns, ds2, T, le, cell)
    ref, plans, fd = _oracle_plan_and_fd(Cs, lambda x, d: (le + x * d.get("le", 0.0))[None, None], ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, thr,
                                         max_nb, dirs, chunk)
    for tile in (0, 4 if (S + L) % 2 == 0 else 16):  # every case through the one-lane body and ONE tile size of the second body (CPU suite time)
        if tile:
            monkeypatch.setenv("XT_EMUL_THG2", str(tile))
        ll, llg, totg, g, plan = E.run_th_grad(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F,
                                               min_len, thr, max_nb, [d[1] for d in dirs], waves=waves, chunk=chunk, capE=512, TT=8, threads=64, nblocks=2)
        # the plan the gradient body followed is the reference algorithm's (index work, exact) ...
        for c, tr in enumerate(plans):
            for i, groups in enumerate(tr):
                got = sorted(tuple(int(v) for v in m) for m in plan[c][i + 2])
                assert got == sorted(tuple(int(v) for v in gg) for gg in groups), (c, i)
        # ... its value is the apply body's and the oracle's, and its gradient the derivative of the oracle at that plan
        assert np.abs(llg - ll).max() < 1e-11 and np.abs(llg - ref).max() < 1e-10 and abs(totg - ref.sum()) < 1e-12 * abs(totg), tile
        rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
        assert rel.max() < 1e-6, (tile, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6])


def test_emulated_frozen_plan_gradient_affine_per_peak_errors(monkeypatch):
    """Per-peak localisation errors with the affine correction (slope / offset, tracking.py:928-930): the two parameters' derivatives; the
    variant with the accumulator rows in the scratch region (what the launcher takes for 3 states and more)."""
    monkeypatch.setenv("XT_EMUL_THG_ROWS_GLOBAL", "1")
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    from oracle import oracle_th as OT
    S, ns, F, L, N, D = 2, 1, 4, 9, 24, 2
    Ds, T, Fs = _model(S, 31)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=5, dims=D)
    rng = np.random.default_rng(4)
    sig = rng.uniform(0.015, 0.03, (N, L, 1))
    slope, offset = 1.1, 0.002
    ds2, cell, pBL, min_len, thr, max_nb = 2 * Ds * 0.02, [1.0], 0.1, 3, 0.2, 120
    dirs = [dict(slope=1.0), dict(offset=1.0), dict(pBL=1.0)]
    ll, llg, totg, g, plan = E.run_th_grad(Cs, sig, np.sqrt(ds2), Fs, T, pBL, 1, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len, thr, max_nb,
                                           dirs, chunk=N, capE=256, TT=8, threads=64, nblocks=2, slope=slope, offset=offset)
    monkeypatch.setenv("XT_EMUL_THG2", "8")  # the (sequence, track)-lane body on the same data
    ll2, llg2, totg2, g2, _ = E.run_th_grad(Cs, sig, np.sqrt(ds2), Fs, T, pBL, 1, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len, thr, max_nb,
                                            dirs, chunk=N, capE=256, TT=8, threads=64, nblocks=2, slope=slope, offset=offset)
    assert np.abs(llg2 - llg).max() < 1e-11 and (np.abs(g2 - g) / np.abs(g)).max() < 1e-9
    tr = []
    eff = lambda sl, of: np.maximum(sig * sl + of, 1e-6)
    ref = OT.proba_cs_th(Cs, eff(slope, offset), np.sqrt(ds2), Fs, T, pBL, 1, cell, ns, F, min_len, thr, max_nb, trace=tr)
    f = lambda sl, of, pb: OT.proba_cs_th(Cs, eff(sl, of), np.sqrt(ds2), Fs, T, pb, 1, cell, ns, F, min_len, thr, max_nb, plan=tr).sum()
    fd = np.array([_richardson(lambda x: f(slope + x, offset, pBL), 1e-4), _richardson(lambda x: f(slope, offset + x, pBL), 1e-6),
                   _richardson(lambda x: f(slope, offset, pBL + x), 1e-4)])
    assert np.abs(llg - ref).max() < 1e-10
    assert (np.abs(g - fd) / np.abs(fd)).max() < 1e-6, (g, fd)
