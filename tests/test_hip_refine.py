"""GPU tests of the position refinement (extrack_refine_positions) through the C ABI: all 50
reference-generated buckets (2-3 states, 3-20 positions, 1-45 tracks, 1-3 dims, different frame_len / threshold / max_nb_states) to
1e-9, a larger bucket against the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_position_refinement_golden():
    from extrack_amd import refined_localization as RL
    meta = json.load(open(os.path.join(GOLDEN, "refine_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_cases.npz"))
    worst_mu = worst_sig = 0.0
    for row in meta:
        pre = "r%04d_" % row["id"]
        g = lambda k: data[pre + k]
        mus, sigs = RL.position_refinement({str(row["L"]): g("Cs")}, row["LocErr"], g("ds"), g("Fs"), g("T"), row["F"], row["threshold"],
                                           row["max_nb_states"])
        dm, dsg = np.abs(mus[str(row["L"])] - g("mu")).max(), np.abs(sigs[str(row["L"])] - g("sig")).max()
        assert dm < 1e-9 and dsg < 1e-9, (row, dm, dsg)
        worst_mu, worst_sig = max(worst_mu, dm), max(worst_sig, dsg)
    print("refine cases", len(meta), "worst |d mu|", worst_mu, "worst |d sigma|", worst_sig)


def test_position_refinement_larger_bucket_vs_oracle_and_errors():
    """700 tracks of 25 positions (23 batches of followers spread over the workgroups): first 60 rows against the oracle run on the same
    first-30-pilots bucket... the merge decisions come from the first 30 tracks of the WHOLE bucket, so the oracle must see the whole
    bucket too; refined positions must lie closer to the simulated true positions than the raw localisations do."""
    from extrack_amd import refined_localization as RL
    from oracle import oracle_refine as OR
    rng = np.random.default_rng(3)
    N, L, S = 700, 25, 2
    ds = np.array([0.01, 0.09])
    Tm = np.array([[0.93, 0.07], [0.12, 0.88]])
    Fs = np.array([0.55, 0.45])
    st = np.zeros((N, L), int)
    st[:, 0] = rng.random(N) > Fs[0]
    for k in range(1, L):
        st[:, k] = np.where(rng.random(N) < Tm[st[:, k - 1], 0], 0, 1)
    truth = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[st][:, :, None], 1)
    Cs = truth + rng.normal(0, 0.03, (N, L, 2))
    mus, sigs = RL.position_refinement({str(L): Cs}, 0.03, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    ref_mu, ref_sig = OR.position_refinement({str(L): Cs}, 0.03, ds, Fs, Tm, 6, 0.1, 100)
    assert np.abs(mus[str(L)] - ref_mu[str(L)]).max() < 1e-9 and np.abs(sigs[str(L)] - ref_sig[str(L)]).max() < 1e-9
    # the same bucket in row blocks of a few dozen tracks (record memory bounded by EXTRACK_REFINE_BUDGET_MB): every block re-walks the
    # pilots' plan, the refined positions do not depend on the blocking
    os.environ["EXTRACK_REFINE_BUDGET_MB"] = "1"
    try:
        mus_b, sigs_b = RL.position_refinement({str(L): Cs}, 0.03, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    finally:
        del os.environ["EXTRACK_REFINE_BUDGET_MB"]
    assert np.array_equal(mus_b[str(L)], mus[str(L)]) and np.array_equal(sigs_b[str(L)], sigs[str(L)])
    raw = np.sqrt(((Cs - truth) ** 2).mean())
    ref = np.sqrt(((mus[str(L)] - truth) ** 2).mean())
    assert ref < 0.9 * raw, (raw, ref)
    assert sigs[str(L)].shape == (N, L) and np.all(sigs[str(L)] > 0) and np.all(sigs[str(L)] < 0.03)
    # a dict of per-peak errors that are all equal: where the reference's mirrored pairing cannot matter the result is the global-error one
    mus_pp, sigs_pp = RL.position_refinement({str(L): Cs}, {str(L): np.full((N, L, 1), 0.03)}, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    assert np.abs(mus_pp[str(L)] - mus[str(L)]).max() < 1e-12 and np.abs(sigs_pp[str(L)] - sigs[str(L)]).max() < 1e-12
    with pytest.raises(ValueError):
        RL.position_refinement({str(L): Cs}, [0.03, 0.03], ds, Fs, Tm)  # per-dimension errors: the reference's reshapes break on them
    # two-position tracks (the reference handles them: both positions are end positions); a mixed dataset incl. a one-track bucket
    short = {"2": Cs[:40, :2], "3": Cs[40:41, :3], "5": Cs[41:75, :5]}
    mus, sigs = RL.position_refinement(short, 0.03, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    ref_mu, ref_sig = OR.position_refinement(short, 0.03, ds, Fs, Tm, 6, 0.1, 100)
    for k in short:
        assert np.abs(mus[k] - ref_mu[k]).max() < 1e-9 and np.abs(sigs[k] - ref_sig[k]).max() < 1e-9, k
    with pytest.raises(Exception):
        RL.position_refinement({"1": Cs[:, :1]}, 0.03, ds, Fs, Tm)


def test_position_refinement_per_peak_errors_golden():
    """Per-peak localisation errors {len: sigma[N, len, 1]} (round 4): all 50 reference-generated buckets (2 - 16 positions, 1 - 45 tracks -
    more than the 30 pilots -, 1 - 3 dims) to 1e-9, the reference's pairing of errors and positions included; shapes the reference's
    reshapes refuse are refused."""
    from extrack_amd import refined_localization as RL
    meta = json.load(open(os.path.join(GOLDEN, "refine_pp_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_pp_cases.npz"))
    worst_mu = worst_sig = 0.0
    for row in meta:
        pre = "p%04d_" % row["id"]
        g = lambda k: data[pre + k]
        key = str(row["L"])
        mus, sigs = RL.position_refinement({key: g("Cs")}, {key: g("sigma")}, g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
        dm, dsg = np.abs(mus[key] - g("mu")).max(), np.abs(sigs[key] - g("sig")).max()
        assert dm < 1e-9 and dsg < 1e-9, (row, dm, dsg)
        worst_mu, worst_sig = max(worst_mu, dm), max(worst_sig, dsg)
    assert len(meta) == 50
    print("per-peak refine cases", len(meta), "worst |d mu|", worst_mu, "worst |d sigma|", worst_sig)
    Cs, sg = data["p0001_Cs"], data["p0001_sigma"]
    with pytest.raises(ValueError):
        RL.position_refinement({str(Cs.shape[1]): Cs}, {str(Cs.shape[1]): np.repeat(sg, Cs.shape[2], 2)}, data["p0001_ds"], data["p0001_Fs"], data["p0001_T"])


def test_position_refinement_per_peak_larger_bucket_in_row_blocks():
    """600 tracks x 20 with per-peak errors against the oracle, also cut into row blocks (the sigma rows must follow the blocks)."""
    from extrack_amd import refined_localization as RL
    from oracle import oracle_refine as OR
    rng = np.random.default_rng(8)
    N, L = 600, 20
    ds, Tm, Fs = np.array([0.01, 0.09]), np.array([[0.93, 0.07], [0.12, 0.88]]), np.array([0.55, 0.45])
    sig = rng.uniform(0.015, 0.04, (N, L, 1))
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, 2, (N, L, 1))], 1) + rng.normal(0, 1, (N, L, 2)) * sig
    key = str(L)
    mus, sigs = RL.position_refinement({key: Cs}, {key: sig}, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    rm, rs = OR.position_refinement({key: Cs}, {key: sig}, ds, Fs, Tm, 6, 0.1, 100)
    assert np.abs(mus[key] - rm[key]).max() < 1e-9 and np.abs(sigs[key] - rs[key]).max() < 1e-9
    os.environ["EXTRACK_REFINE_BUDGET_MB"] = "1"
    try:
        mb, sb = RL.position_refinement({key: Cs}, {key: sig}, ds, Fs, Tm, frame_len=6, threshold=0.1, max_nb_states=100)
    finally:
        del os.environ["EXTRACK_REFINE_BUDGET_MB"]
    assert np.array_equal(mb[key], mus[key]) and np.array_equal(sb[key], sigs[key])


def test_get_pos_pdf_components_golden():
    """get_pos_PDF's return values through extrack_refine_pos_pdf (round 4): 40 reference-generated buckets (2 - 11 positions, 1 - 36 tracks,
    global and per-peak errors): the number of components of every position exactly, means / stds to 1e-9, log-weights to 1e-8; the
    read-out of the components reproduces position_refinement."""
    from extrack_amd import refined_localization as RL
    meta = json.load(open(os.path.join(GOLDEN, "refine_pdf_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_pdf_cases.npz"))
    worst = 0.0
    for row in meta:
        pre = "d%04d_" % row["id"]
        g = lambda k: data[pre + k]
        means, stds, wts = RL.get_pos_PDF(g("Cs"), g("sigma"), g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
        assert [w.shape[1] for w in wts] == list(g("counts")), row
        assert all(s.shape == w.shape + (1,) for s, w in zip(stds, wts)) and all(m.shape[:2] == w.shape for m, w in zip(means, wts))
        dm = np.abs(np.concatenate(means, 1) - g("means")).max()
        dsg = np.abs(np.concatenate([s[:, :, 0] for s in stds], 1) - g("stds")).max()
        dw = np.abs(np.concatenate(wts, 1) - g("logw")).max()
        assert dm < 1e-9 and dsg < 1e-9 and dw < 1e-8, (row, dm, dsg, dw)
        worst = max(worst, dm, dsg, dw)
    assert len(meta) == 40
    print("get_pos_PDF cases", len(meta), "worst difference", worst)
    # read-out of the components (refined_localization.py:329-337) = position_refinement on the same bucket
    row = meta[12]
    g = lambda k: data["d%04d_" % row["id"] + k]
    key = str(row["L"])
    le = {key: g("sigma")} if row["per_peak"] else float(g("sigma").ravel()[0])
    mus, sigs = RL.position_refinement({key: g("Cs")}, le, g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
    means, stds, wts = RL.get_pos_PDF(g("Cs"), g("sigma"), g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
    for k, (pm, ps, pw) in enumerate(zip(means, stds, wts)):
        P = np.exp(pw - pw.max(1, keepdims=True))
        assert np.abs((P[:, :, None] * pm).sum(1) / P.sum(1)[:, None] - mus[key][:, k]).max() < 1e-12
        assert np.abs(((P * ps[:, :, 0] ** 2).sum(1) / P.sum(1)) ** 0.5 - sigs[key][:, k]).max() < 1e-12
    # a bucket whose records do not fit one row block is refused (the components of a large bucket are not meant to leave the GPU)
    from extrack_amd import _lib
    os.environ["EXTRACK_REFINE_BUDGET_MB"] = "1"
    try:
        rng = np.random.default_rng(0)
        with pytest.raises(_lib.ExtrackError):
            RL.get_pos_PDF(np.cumsum(rng.normal(0, 0.05, (3000, 30, 2)), 1), 0.02, [0.01, 0.09], [0.5, 0.5], np.array([[0.9, 0.1], [0.1, 0.9]]), 6, 0.1, 100)
    finally:
        del os.environ["EXTRACK_REFINE_BUDGET_MB"]
