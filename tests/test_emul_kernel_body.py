"""The HIP kernel bodies (extrack_amd/csrc/xt_kernel.h generic path, xt_fast2.h two-state fast path), compiled for
the host and run on CPU threads (tests/emul), against the golden vectors generated from the reference.  This is what
lets the index logic and the extended-range arithmetic be checked on a box without a GPU; the GPU run of the very
same source is tests/test_hip_parity.py."""
import shutil

import numpy as np
import pytest

from conftest import case_inputs
from oracle import oracle_np as O

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _run(row, x, preds, **kw):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    ps = O.p_stay_table(x["ds"], row["S"], row["ns"], row["cell_dims"])
    return E.run(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], ps, row["ns"], row["F"], row["min_len"],
                 preds=preds, **kw)


def test_generic_body_on_golden_subset(kernel_cases, monkeypatch):
    monkeypatch.setenv("XT_EMUL_GENERIC", "1")
    meta, data = kernel_cases
    worst = worstp = 0.0
    n = 0
    for row in meta:
        if row["S"] ** row["F"] > 300 or row["id"] % 5:
            continue
        x = case_inputs(row, data)
        pr = row["ns"] == 1 and row["S"] <= 6
        ll, p, tot, _ = _run(row, x, pr)
        worst = max(worst, np.abs(ll - x["LPC"]).max())
        assert abs(tot - ll.sum()) < 1e-9
        if pr:
            worstp = max(worstp, np.abs(p - x["preds"]).max())
        n += 1
    assert n > 150 and worst < 1e-10 and worstp < 1e-9, (n, worst, worstp)


@pytest.mark.parametrize("reg2", [False, True])
@pytest.mark.parametrize("guarded", [False, True])
def test_fast2_body_on_golden_two_state_cases(kernel_cases, guarded, reg2, monkeypatch):
    """guarded: the fully guarded steps of the fast path (per-step normalisation of every weight, zero handling) instead of the lazy /
    zero-free ones that well-scaled models take.  reg2: the register-resident body (xt_reg2.h, the product's default 2-state kernel:
    state in VGPRs, lane exchanges) instead of the LDS-resident one (xt_fast2.h)."""
    monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    if reg2:
        monkeypatch.setenv("XT_EMUL_REG2", "1")
    else:
        monkeypatch.delenv("XT_EMUL_REG2", raising=False)
    if guarded:
        monkeypatch.setenv("XT_EMUL_GUARDED", "1")
    else:
        monkeypatch.delenv("XT_EMUL_GUARDED", raising=False)
    meta, data = kernel_cases
    worst = 0.0
    n = 0
    for row in meta:
        if not (row["S"] == 2 and row["ns"] == 1 and row["F"] in (4, 6)) or row["id"] % (4 if reg2 else 2):
            continue
        x = case_inputs(row, data)
        ll, _, tot, info = _run(row, x, False, nblocks=1)
        assert info[1] == 256  # fast path geometry: 4 independent waves per block
        worst = max(worst, np.abs(ll - x["LPC"]).max())
        assert abs(tot - ll.sum()) < 1e-9
        n += 1
    assert n > (30 if reg2 else 60) and worst < 1e-10, (n, worst)


@pytest.mark.parametrize("F,L,N,reg2", [(5, 40, 11, 0), (7, 70, 5, 0), (6, 33, 9, 0), (4, 65, 17, 0), (5, 40, 11, 1), (7, 70, 5, 1), (6, 33, 9, 1), (4, 65, 17, 1)])
def test_fast2_multi_chunk_tracks(F, L, N, reg2, monkeypatch):
    """Tracks longer than one 32-position staging chunk, partial last batch, several blocks (reg2: register-resident body)."""
    monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    if reg2:
        monkeypatch.setenv("XT_EMUL_REG2", "1")
    else:
        monkeypatch.delenv("XT_EMUL_REG2", raising=False)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    rng = np.random.default_rng(F * 100 + L)
    Cs = np.cumsum(rng.normal(0, 0.05, (N, L, 2)), 1)
    ds, Fs, T = np.array([0.01, 0.1]), np.array([.4, .6]), np.array([[.9, .1], [.2, .8]])
    LE = np.array([[[0.02]]])
    ps = O.p_stay_table(ds, 2, 1, [1.0])
    ref = O.proba_cs(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], 1, F, 3)
    ll, _, tot, _ = E.run(Cs, LE, ds, Fs, T, 0.1, 1, ps, 1, F, 3, nblocks=2)
    assert np.abs(ll - ref).max() < 1e-10
    assert abs(tot - ref.sum()) < 1e-9


def test_fast2_tiny_and_zero_transition_probabilities(monkeypatch):
    """Models outside the 'well-scaled' bounds of the 2-state fast path take its guarded steps (every stored weight normalised, zero
    weights handled): transition probabilities of 1e-300 / 1e-200 against the oracle (found in round 2: anything below ~1e-100 used to
    give NaN, the step's W^3-sized products underflowed), an exactly zero one against the 1e-300 result."""
    monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    rng = np.random.default_rng(4)
    Cs = np.cumsum(rng.normal(0, 0.05, (9, 14, 2)), 1)
    ds, Fs = np.array([0.002, 0.1]), np.array([.3, .7])
    LE = np.array([[[0.02]]])
    ps = O.p_stay_table(ds, 2, 1, [1.0])
    got = {}
    for eps in (0.0, 1e-300, 1e-200, 1e-25, 1e-3):
        T = np.array([[1.0 - eps, eps], [0.15, 0.85]])
        got[eps] = E.run(Cs, LE, ds, Fs, T, 0.1, 1, ps, 1, 6, 3, nblocks=1)[0]
        if eps > 0:
            assert np.abs(got[eps] - O.proba_cs(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], 1, 6, 3)).max() < 1e-10, eps
    assert np.all(np.isfinite(got[0.0])) and np.abs(got[0.0] - got[1e-300]).max() < 1e-10


@pytest.mark.parametrize("generic", [False, True])
def test_absurd_jump_stays_a_tiny_weight(generic, monkeypatch):
    """A jump of thousands of localisation errors in one frame: the Gaussian exponent leaves the range of the table-driven exp.  The
    sequence's weight must clamp to (practically) zero - not wrap around the int32 exponent into a huge one (found in round 2: the
    clamp was -3e7 while 64 x / ln2 only fits int32 down to -2.3e7).  Jumps inside the range stay exact."""
    if generic:
        monkeypatch.setenv("XT_EMUL_GENERIC", "1")
    else:
        monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    rng = np.random.default_rng(8)
    N, L = 6, 12
    Cs = np.cumsum(rng.normal(0, 0.05, (N, L, 2)), 1)
    Cs[1, 6:] += 750.0   # Gaussian exponent ~ -2.6e7 for the mobile state: beyond the clamp (and in the range that used to wrap)
    Cs[2, 6:] += 60.0    # ~ -1.7e5: inside
    ds, Fs, T = np.array([0.01, 0.1]), np.array([.4, .6]), np.array([[.9, .1], [.2, .8]])
    LE = np.array([[[0.02]]])
    ps = O.p_stay_table(ds, 2, 1, [1.0])
    ref = O.proba_cs(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], 1, 6, 3)
    ll, _, tot, _ = E.run(Cs, LE, ds, Fs, T, 0.1, 1, ps, 1, 6, 3, nblocks=1)
    ok = np.array([0, 2, 3, 4, 5])
    assert (np.abs(ll[ok] - ref[ok]) < 1e-10 + 1e-12 * np.abs(ref[ok])).all(), (ll, ref)
    assert np.isfinite(ll[1]) and ll[1] < -7e6 and ref[1] < -7e6, (ll[1], ref[1])   # clamped: hugely negative, not garbage


@pytest.mark.parametrize("S,ns,F,L,N,pred", [(3, 1, 3, 40, 7, True), (2, 1, 4, 70, 9, True), (2, 2, 3, 66, 5, False), (3, 1, 4, 33, 4, True)])
def test_generic_body_multi_chunk_and_per_peak(S, ns, F, L, N, pred, monkeypatch):
    """General kernel body with tracks longer than one 32-position staging chunk, per-peak localisation errors,
    likelihood and posteriors, against the oracle."""
    monkeypatch.setenv("XT_EMUL_GENERIC", "1")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    rng = np.random.default_rng(S * 1000 + L)
    ds = np.sort(rng.uniform(0.01, 0.15, S))
    Fs = np.full(S, 1 / S)
    T = np.full((S, S), 0.08)
    T[np.arange(S), np.arange(S)] = 1 - 0.08 * (S - 1)
    Cs = np.cumsum(rng.normal(0, 0.05, (N, L, 2)), 1)
    LE = rng.uniform(0.01, 0.03, (N, L, 2))
    ps = O.p_stay_table(ds, S, ns, [1.0])
    LPr, pr = O.p_cs_inter_bound_stats(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 1 if pred else 0, 3)
    mx = LPr.max(1, keepdims=True)
    ref = np.log(np.exp(LPr - mx).sum(1)) + mx[:, 0]
    ll, p, tot, _ = E.run(Cs, LE, ds, Fs, T, 0.1, 1, ps, ns, F, 3, preds=pred, nblocks=2)
    assert np.abs(ll - ref).max() < 1e-10
    if pred:
        assert np.abs(p - pr).max() < 1e-9


def test_entry_parallel_body_on_multi_substep_cases(kernel_cases, monkeypatch):
    """nb_substeps >= 2 goes through the entry-parallel kernel body (xt_entry.h)."""
    monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    meta, data = kernel_cases
    worst, n = 0.0, 0
    for row in meta:
        if row["ns"] < 2 or row["id"] % 9:
            continue
        x = case_inputs(row, data)
        ll, _, tot, _ = _run(row, x, False, nblocks=2)
        worst = max(worst, np.abs(ll - x["LPC"]).max())
        assert abs(tot - ll.sum()) < 1e-9
        n += 1
    assert n > 40 and worst < 1e-10, (n, worst)


@pytest.mark.parametrize("S,ns,F", [(2, 1, 6), (3, 1, 3), (2, 2, 3)])
def test_multi_bucket_single_launch(S, ns, F, monkeypatch):
    """Several length buckets served by ONE launch through the bucket-descriptor table (all three kernel bodies): every
    block must pick its own bucket's length, isBL flag and track range."""
    monkeypatch.delenv("XT_EMUL_GENERIC", raising=False)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    rng = np.random.default_rng(S * 10 + F)
    ds = np.sort(rng.uniform(0.01, 0.15, S))
    Fs = np.full(S, 1 / S)
    T = np.full((S, S), 0.08)
    T[np.arange(S), np.arange(S)] = 1 - 0.08 * (S - 1)
    lens = {2: 3, 5: 11, 9: 7, 12: 5}
    buckets = [np.cumsum(rng.normal(0, 0.05, (n, L, 2)), 1) for L, n in lens.items()]
    ps = O.p_stay_table(ds, S, ns, [1.0])
    LE = np.array([[[0.02]]])
    ref = [O.proba_cs(b, LE, ds, Fs, T, 0.1, 0 if b.shape[1] == 12 else 1, [1.0], ns, F, 2) for b in buckets]
    outs, tot = E.run_multi(buckets, [0.02], ds, Fs, T, 0.1, ps, ns, F, 2, 12, [1, 3, 2, 2])
    for o, r in zip(outs, ref):
        assert np.abs(o - r).max() < 1e-10
    assert abs(tot - sum(r.sum() for r in ref)) < 1e-9


def test_sequence_matrix_on_golden_cases(kernel_cases):
    """P_Cs_inter_bound_stats' per-sequence matrix LP[N, nB] (the reference's first return value, extrack/tracking.py:318): the general
    kernel body's raw per-(sequence, new digits) output mapped to the reference's column order by csrc/xt_seqmat.h, against the matrices
    the reference produced (stored in the golden cases with nB <= 256): 2-4 states, nb_substeps 1-2, isBL 0 / 1, tracks shorter and
    longer than the window (the first 25 distinct configurations here; all 670 matrices on the GPU).  Entries are compared where the reference is finite; -inf / underflowed entries must be <= -700."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    meta, data = kernel_cases
    n, worst, seen = 0, 0.0, set()
    for row in meta:
        x = case_inputs(row, data)
        if x["LP"] is None or row["S"] ** row["F"] > 300:
            continue
        key = (row["S"], row["ns"], row["F"], row["isBL"], x["Cs"].shape[1], x["Cs"].shape[2], x["LE"].shape[1:])
        if key in seen:
            continue
        seen.add(key)
        ps = O.p_stay_table(x["ds"], row["S"], row["ns"], row["cell_dims"])
        lp = E.run_seq_matrix(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], ps, row["ns"], row["F"], row["min_len"])
        assert lp.shape == x["LP"].shape, (row, lp.shape, x["LP"].shape)
        fin = np.isfinite(x["LP"]) & (x["LP"] > -650)
        worst = max(worst, np.abs(lp[fin] - x["LP"][fin]).max())
        assert np.all(lp[~fin] < -600)
        n += 1
        if n >= 25:  # the GPU test (tests/test_hip_parity.py) runs all 670 reference matrices
            break
    assert n >= 25 and worst < 1e-9, (n, worst)


@pytest.mark.parametrize("S,ns,F,L,N,D,K,isBL,preds", [(2, 1, 4, 9, 70, 2, 1, 1, False), (3, 1, 3, 8, 66, 2, 2, 0, True), (2, 2, 4, 7, 10, 1, 1, 1, False),
                                                        (4, 1, 3, 6, 9, 3, 3, 1, True), (2, 1, 5, 2, 5, 2, 1, 1, True), (3, 1, 4, 3, 5, 2, 1, 0, True),
                                                        (2, 1, 11, 14, 3, 2, 1, 1, False), (5, 1, 5, 7, 2, 2, 1, 1, True)])
def test_emulated_global_state_body_for_big_models(S, ns, F, L, N, D, K, isBL, preds, monkeypatch):
    """csrc/xt_big.h (one lane per track, sequence state in global memory: the models whose state does not fit a workgroup - here also forced
    on small ones, XT_EMUL_BIG) on CPU threads against the pinned oracle: per-track LL 1e-10, posteriors 1e-9.  (2, 1, 11): 2048 groups per
    track, (5, 1, 5): 3125 sequences - both refused by the LDS kernels."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    monkeypatch.setenv("XT_EMUL_BIG", "1")
    rng = np.random.default_rng(S * 100 + F)
    Ds = np.sort(rng.uniform(0.01, 0.3, S))
    Ds[0] = 0.001
    T = np.full((S, S), 0.05) + rng.uniform(0, 0.03, (S, S))
    T[np.arange(S), np.arange(S)] = 0
    T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    Fs = rng.dirichlet(np.ones(S) * 3)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F + L, dims=D)
    ds, cell, pBL, min_len = np.sqrt(2 * Ds * 0.02), [1.0], 0.1, 3
    le = np.array([0.02, 0.025, 0.03][:K])[None, None]
    ll, pr, tot, info = E.run(Cs, le, ds, Fs, T, pBL, isBL, O.p_stay_table(ds, S, ns, cell), ns, F, min_len, preds=preds, nblocks=2)
    ref = O.proba_cs(Cs, le, ds, Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)
    if preds:
        refp = O.p_cs_inter_bound_stats(Cs, le, ds, Fs, T, pBL, isBL, cell, ns, F, 1, min_len)[1]
        assert np.abs(pr - refp).max() < 1e-9
