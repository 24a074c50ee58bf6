"""The threshold-fusion kernel bodies (extrack_amd/csrc/xt_th.h: plan kernel, the three apply-kernel variants), compiled for the
host and run on CPU threads (tests/emul), against the reference-generated golden vectors and the pinned oracle
(oracle/oracle_th.py).  The merge groups (index work) must be identical, log-likelihoods within 1e-10.  The GPU run of the
same source is tests/test_hip_th_parity.py."""
import json
import os
import shutil
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle_th as OT
from oracle.oracle_np import p_stay_table

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _emul():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emul"))
    import run_emul as E
    return E


def test_th_bodies_on_golden_subset():
    E = _emul()
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases.npz"))
    rows = [r for r in meta if not r["do_preds"]][::14]
    worst = 0.0
    for row in rows:
        pre = "t%04d_" % row["id"]
        Cs, LE, ds, Fs, T = [data[pre + k] for k in ("Cs", "LE", "ds", "Fs", "T")]
        ps = p_stay_table(ds, len(ds), row["ns"], row["cell_dims"])
        tr = []
        OT.p_cs_inter_bound_stats_th(Cs, LE, ds, Fs, T, row["pBL"], row["isBL"], row["cell_dims"], row["ns"], row["F"], 0, row["min_len"],
                                     row["threshold"], row["max_nb_states"], trace=tr)
        ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, row["pBL"], row["isBL"], ps, row["ns"], row["F"], row["min_len"],
                                              row["threshold"], row["max_nb_states"], chunk=len(Cs), capE=256, TT=8, threads=64, nblocks=2)
        for i, t in enumerate(range(2, row["L"] - 1)):
            assert [list(g) for g in tr[i]] == [list(g) for g in plan[0][t]], (row, t)
        worst = max(worst, np.abs(ll - data[pre + "LPC"]).max())
        assert abs(tot - ll.sum()) < 1e-9
    assert len(rows) >= 13 and worst < 1e-10, (len(rows), worst)


@pytest.mark.parametrize("threshold,max_nb,threads,staged", [(0.1, 100, 192, True)])
def test_th_plan_many_sequences_wrapped_classes(threshold, max_nb, threads, staged, monkeypatch):
    """More than 128 expanded sequences per step with 3 states: the reference's int8 index wrap makes the history classes irregular;
    the plan kernel then walks per-class candidate lists.  Several wavefronts, global workspace with / without the LDS staging copy.
    Groups identical to the oracle's."""
    if staged:
        for k in ("STP", "STE"):
            monkeypatch.setenv("XT_EMUL_TH_" + k, "600")
    E = _emul()
    rng = np.random.default_rng(5)
    S, ns, F, L, N = 3, 1, 6, 9, 36
    ds = np.array([0.004, 0.03, 0.11])
    Fs = np.array([0.3, 0.3, 0.4])
    T = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2))
    LE = np.full((1, 1, 1), 0.02)
    ps = p_stay_table(ds, S, ns, [1.0])
    tr = []
    ref = OT.proba_cs_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 5, threshold, max_nb)
    OT.p_cs_inter_bound_stats_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 0, 5, threshold, max_nb, trace=tr)
    ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, 0.1, 1, ps, ns, F, 5, threshold, max_nb, chunk=N, capE=1024, TT=8, threads=threads, nblocks=1)
    assert status[:, 0].max() == 0
    assert status[0, 1] > 128, status  # expanded sequences: the shared-row path and the wrapped classes were exercised
    for i, t in enumerate(range(2, L - 1)):
        assert [list(g) for g in tr[i]] == [list(g) for g in plan[0][t]], t
    assert np.abs(ll - ref).max() < 1e-10


@pytest.mark.parametrize("variant", ["general", "streamed", "direct", "uniform", "uniform_single", "general_single", "general_single_streamed", "general_single_direct",
                                     "lds_workspace"])
def test_th_apply_variants_chunked(variant, monkeypatch):
    """Two chunks (the second ragged), 3 states, per-peak errors: every apply-kernel variant and the LDS-resident plan workspace."""
    E = _emul()
    rng = np.random.default_rng(17)
    S, ns, F, L, N, chunk = 3, 1, 4, 9, 100, 70
    ds = np.sort(rng.uniform(0.005, 0.2, S))
    Fs = rng.dirichlet(np.ones(S) * 2)
    T = rng.uniform(0.02, 0.25, (S, S))
    T[np.arange(S), np.arange(S)] = 0
    T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2))
    LE = rng.uniform(0.01, 0.04, (N, L, 2))
    ps = p_stay_table(ds, S, ns, [0.8])
    ref = np.concatenate([OT.proba_cs_th(Cs[a:a + chunk], LE[a:a + chunk], ds, Fs, T, 0.07, 1, [0.8], ns, F, 3, 0.2, 30) for a in range(0, N, chunk)])
    kw = dict(chunk=chunk, capE=128, TT=8, threads=128, nblocks=2)
    if variant in ("streamed", "direct"):
        kw["nblocks"] = -2
        if variant == "direct":  # member lists read from global memory (what the launcher takes when one step's list would fill the LDS)
            monkeypatch.setenv("XT_EMUL_TH_DIRECT", "1")
    elif variant == "uniform":
        kw.update(TT=64, threads=256)
    elif variant == "uniform_single":
        kw.update(TT=64, threads=256)
        monkeypatch.setenv("XT_EMUL_TH_SINGLE", "1")
    elif variant in ("general_single", "general_single_streamed", "general_single_direct"):
        kw.update(TT=8, threads=128, nblocks=2 if variant == "general_single" else -2)
        monkeypatch.setenv("XT_EMUL_TH_SINGLE", "1")
        if variant.endswith("direct"):
            monkeypatch.setenv("XT_EMUL_TH_DIRECT", "1")
    elif variant == "lds_workspace":
        monkeypatch.setenv("XT_EMUL_TH_WSP", "24")
        monkeypatch.setenv("XT_EMUL_TH_WSE", "60")
    ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, 0.07, 1, ps, ns, F, 3, 0.2, 30, **kw)
    assert status[:, 0].max() == 0
    assert np.abs(ll - ref).max() < 1e-10
    assert abs(tot - ref.sum()) < 1e-9


def test_th_posteriors_body_on_golden_subset():
    """Prediction mode of the plan body (per-track state histories, weighted merges, posterior read-out)."""
    E = _emul()
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases.npz"))
    rows = [r for r in meta if r["do_preds"] and r["N"] <= 30][::9]
    worst = 0.0
    for row in rows:
        pre = "t%04d_" % row["id"]
        Cs, LE, ds, Fs, T = [data[pre + k] for k in ("Cs", "LE", "ds", "Fs", "T")]
        ps = p_stay_table(ds, len(ds), 1, row["cell_dims"])
        pr = E.run_th_predict(Cs, LE, ds, Fs, T, row["pBL"], row["isBL"], ps, row["F"], row["min_len"], row["threshold"], row["max_nb_states"],
                              chunk=len(Cs), capE=256, threads=64, nblocks=2)
        worst = max(worst, np.abs(pr - data[pre + "preds"]).max())
    assert len(rows) >= 8 and worst < 1e-9, (len(rows), worst)


def test_th_posteriors_body_chunks_beyond_the_pilots():
    """Reference fixtures with 40 / 60 tracks per chunk (predict_Bs with nb_max > 30): the first 30 tracks decide the merges, the
    others replay the plan with their own merge weights (tracking.py:676-691, 703-741)."""
    E = _emul()
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases.npz"))
    rows = [r for r in meta if r["do_preds"] and r["N"] > 30 and r["L"] <= 12][::7]
    worst = 0.0
    for row in rows:
        pre = "t%04d_" % row["id"]
        Cs, LE, ds, Fs, T = [data[pre + k] for k in ("Cs", "LE", "ds", "Fs", "T")]
        ps = p_stay_table(ds, len(ds), 1, row["cell_dims"])
        pr = E.run_th_predict(Cs, LE, ds, Fs, T, row["pBL"], row["isBL"], ps, row["F"], row["min_len"], row["threshold"], row["max_nb_states"],
                              chunk=len(Cs), capE=256, threads=64, nblocks=2)
        d = np.abs(pr - data[pre + "preds"]).max()
        assert d < 1e-9, (row, d)
        worst = max(worst, d)
    assert len(rows) >= 8
    print("chunks beyond the pilots:", len(rows), "worst", worst)


@pytest.mark.parametrize("TT,threads", [(8, 128), (64, 256)])
def test_th_multi_bucket_launch(TT, threads):
    """Three buckets of different lengths (one with a single ragged chunk, one shorter than the merge horizon) served by ONE plan
    launch and ONE apply launch through the bucket-descriptor table; isBL per bucket from the dataset's max length."""
    E = _emul()
    rng = np.random.default_rng(23)
    S, ns, F, chunk = 2, 1, 5, 40
    ds = np.array([0.012, 0.11])
    Fs = np.array([0.45, 0.55])
    T = np.array([[0.9, 0.1], [0.15, 0.85]])
    buckets = []
    for L, N in ((3, 55), (8, 100), (13, 37)):
        buckets.append(np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2)))
    ps = p_stay_table(ds, S, ns, [0.8])
    outs, tot = E.run_th_multi(buckets, [0.02], ds, Fs, T, 0.07, ps, ns, F, 3, 13, 0.2, 30, chunk, TT=TT, threads=threads)
    LE = np.array([[[0.02]]])
    for b, o in zip(buckets, outs):
        isBL = 0 if b.shape[1] == 13 else 1
        ref = np.concatenate([OT.proba_cs_th(b[a:a + chunk], LE, ds, Fs, T, 0.07, isBL, [0.8], ns, F, 3, 0.2, 30) for a in range(0, len(b), chunk)])
        assert np.abs(o - ref).max() < 1e-10
    assert abs(tot - sum(o.sum() for o in outs)) < 1e-9


def test_th_bodies_on_extra_golden_subset():
    """A few small cases of the second reference fixture (1-D / 3-D, per-dimension errors, nb_substeps 3, 5 states)."""
    E = _emul()
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases_extra.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases_extra.npz"))
    rows = [r for r in meta if r["N"] <= 7 and r["L"] <= 7 and r["nB"] <= 60]
    worst = 0.0
    for row in rows:
        pre = "x%04d_" % row["id"]
        Cs, LE, ds, Fs, T = [data[pre + k] for k in ("Cs", "LE", "ds", "Fs", "T")]
        ps = p_stay_table(ds, len(ds), row["ns"], row["cell_dims"])
        ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, row["pBL"], row["isBL"], ps, row["ns"], row["F"], row["min_len"],
                                              row["threshold"], row["max_nb_states"], chunk=len(Cs), capE=1024, TT=8, threads=64, nblocks=2)
        worst = max(worst, np.abs(ll - data[pre + "LPC"]).max())
    assert len(rows) >= 6 and worst < 1e-10, (len(rows), worst)


def test_th_bodies_with_per_track_time_steps():
    """Reference fixtures with dt per track and position (3-D ds): plan + apply (LL) and the prediction mode, incl. chunks beyond the
    30 pilot tracks; the diffusion term of step t is scaled by dt[track, len - t], every chunk has its own field-of-view table."""
    E = _emul()
    info = json.load(open(os.path.join(GOLDEN, "th_dt_cases.json")))
    data = np.load(os.path.join(GOLDEN, "th_dt_cases.npz"))
    rows = [r for r in info["cases"] if r["id"] % 7 == 0 or (r["N"] == 45 and r["id"] % 3 == 0)]
    worst_ll = worst_pr = 0.0
    nll = npr = 0
    for row in rows:
        pre = "d%04d_" % row["id"]
        g = lambda k: data[pre + k]
        Cs, LE, dt, Ds, Fs, T = [np.ascontiguousarray(g(k)) for k in ("Cs", "LE", "dt", "Ds", "Fs", "T")]
        S = len(Ds)
        ds_unit = np.sqrt(2 * Ds)
        ps = np.ascontiguousarray(p_stay_table(np.median(np.sqrt(2 * Ds[None] * dt[:, 0, None]), axis=0), S, row["ns"], row["cell_dims"])[None])
        E.set_th_dt(dt, ps)
        if row["do_preds"]:
            pr = E.run_th_predict(Cs, LE, ds_unit, Fs, T, row["pBL"], row["isBL"], ps[0], row["F"], row["min_len"], row["threshold"],
                                  row["max_nb_states"], chunk=len(Cs), capE=256, threads=64, nblocks=1)
            worst_pr = max(worst_pr, np.abs(pr - g("preds")).max())
            npr += 1
        else:
            ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds_unit, Fs, T, row["pBL"], row["isBL"], ps[0], row["ns"], row["F"], row["min_len"],
                                                  row["threshold"], row["max_nb_states"], chunk=len(Cs), capE=256, TT=8, threads=64, nblocks=1)
            worst_ll = max(worst_ll, np.abs(ll - g("LPC")).max())
            nll += 1
    assert nll >= 10 and npr >= 10 and worst_ll < 1e-10 and worst_pr < 1e-9, (nll, npr, worst_ll, worst_pr)
    print("dt cases: LL", nll, worst_ll, "posteriors", npr, worst_pr)


def test_th_plan_arrays_in_global_workspace(monkeypatch):
    """The per-step plan arrays (member words, newest states, member / group-start lists) at the head of the global workspace instead of
    LDS - what the library does beyond 8192 expanded sequences per step - forced on a 3-state case: groups identical to the oracle's."""
    monkeypatch.setenv("XT_EMUL_TH_PLAN_GLB", "1")
    E = _emul()
    rng = np.random.default_rng(6)
    S, ns, F, L, N = 3, 1, 5, 8, 33
    ds = np.array([0.004, 0.03, 0.11])
    Fs = np.array([0.3, 0.3, 0.4])
    T = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2))
    LE = np.full((1, 1, 1), 0.02)
    ps = p_stay_table(ds, S, ns, [1.0])
    tr = []
    ref = OT.proba_cs_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 5, 0.15, 60)
    OT.p_cs_inter_bound_stats_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 0, 5, 0.15, 60, trace=tr)
    ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, 0.1, 1, ps, ns, F, 5, 0.15, 60, chunk=N, capE=512, TT=8, threads=128, nblocks=1)
    assert status[:, 0].max() == 0
    for i, t in enumerate(range(2, L - 1)):
        assert [list(g) for g in tr[i]] == [list(g) for g in plan[0][t]], t
    assert np.abs(ll - ref).max() < 1e-10


@pytest.mark.skipif(not os.environ.get("XT_SLOW_TESTS"), reason="2 minutes of CPU-thread emulation (16 384 sequences): XT_SLOW_TESTS=1; the GPU suite runs the same case")
def test_th_more_than_8192_expanded_sequences():
    """4 states x 3 substeps (the C5 model): 4^4 = 256 sequences after the first position, 256 x 4^3 = 16 384 expanded at the second - beyond
    the 8192 whose plan arrays fit the LDS (refused until round 4).  Merge groups identical to the oracle's, LL within 1e-10."""
    E = _emul()
    rng = np.random.default_rng(12)
    S, ns, F, L, N = 4, 3, 4, 4, 3
    ds = np.array([0.003, 0.02, 0.06, 0.15])
    Fs = np.array([0.25, 0.3, 0.2, 0.25])
    T = np.full((S, S), 0.02) + np.diag([0.01, 0.0, 0.015, 0.005])
    T[np.arange(S), np.arange(S)] = 0
    T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2))
    LE = np.full((1, 1, 1), 0.02)
    ps = p_stay_table(ds, S, ns, [1.0])
    tr = []
    ref = OT.proba_cs_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 3, 0.3, 120)
    OT.p_cs_inter_bound_stats_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 0, 3, 0.3, 120, trace=tr)
    ll, tot, plan, hdr, status = E.run_th(Cs, LE, ds, Fs, T, 0.1, 1, ps, ns, F, 3, 0.3, 120, chunk=N, capE=16384, TT=1, threads=256, nblocks=1)
    assert status[:, 0].max() == 0 and status[0, 1] == 16384, status
    for i, t in enumerate(range(2, L - 1)):
        assert [list(g) for g in tr[i]] == [list(g) for g in plan[0][t]], t
    assert np.abs(ll - ref).max() < 1e-10
