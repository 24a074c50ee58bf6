"""The position-refinement oracle (oracle/oracle_refine.py) against golden vectors produced by the reference itself
(tests/golden/make_golden_refine.py: extrack/refined_localization.py position_refinement on 50 buckets)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_position_refinement_matches_reference():
    from oracle import oracle_refine as OR
    meta = json.load(open(os.path.join(GOLDEN, "refine_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_cases.npz"))
    worst_mu = worst_sig = 0.0
    for row in meta:
        pre = "r%04d_" % row["id"]
        g = lambda k: data[pre + k]
        mus, sigs = OR.position_refinement({str(row["L"]): g("Cs")}, row["LocErr"], g("ds"), g("Fs"), g("T"), row["F"], row["threshold"],
                                           row["max_nb_states"])
        worst_mu = max(worst_mu, np.abs(mus[str(row["L"])] - g("mu")).max())
        worst_sig = max(worst_sig, np.abs(sigs[str(row["L"])] - g("sig")).max())
    assert worst_mu < 1e-10 and worst_sig < 1e-10, (worst_mu, worst_sig)
    print("refine cases", len(meta), "worst |d mu|", worst_mu, "worst |d sigma|", worst_sig)


def test_position_refinement_per_peak_errors_matches_reference():
    """Per-peak localisation errors {len: sigma[N, len, 1]} as the reference computes them, its pairing of errors and positions in the pass
    "from the future" included (oracle_refine's docstring): 50 reference-generated buckets (2 - 16 positions, 1 - 45 tracks, 1 - 3 dims)."""
    from oracle import oracle_refine as OR
    meta = json.load(open(os.path.join(GOLDEN, "refine_pp_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_pp_cases.npz"))
    worst_mu = worst_sig = 0.0
    for row in meta:
        pre = "p%04d_" % row["id"]
        g = lambda k: data[pre + k]
        key = str(row["L"])
        mus, sigs = OR.position_refinement({key: g("Cs")}, {key: g("sigma")}, g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
        worst_mu = max(worst_mu, np.abs(mus[key] - g("mu")).max())
        worst_sig = max(worst_sig, np.abs(sigs[key] - g("sig")).max())
    assert len(meta) == 50 and worst_mu < 1e-10 and worst_sig < 1e-10, (worst_mu, worst_sig)
    print("per-peak refine cases", len(meta), "worst |d mu|", worst_mu, "worst |d sigma|", worst_sig)


def test_pos_pdf_components_match_reference():
    """get_pos_PDF's own return values (means / stds / log-weights of every mixture component of every position, in the reference's component
    order): 40 reference-generated buckets, global and per-peak errors (tests/golden/make_golden_refine.py pdf)."""
    from oracle import oracle_refine as OR
    meta = json.load(open(os.path.join(GOLDEN, "refine_pdf_cases.json")))
    data = np.load(os.path.join(GOLDEN, "refine_pdf_cases.npz"))
    worst = 0.0
    for row in meta:
        pre = "d%04d_" % row["id"]
        g = lambda k: data[pre + k]
        means, stds, wts = OR.pos_pdf(g("Cs"), g("sigma"), g("ds"), g("Fs"), g("T"), row["F"], row["threshold"], row["max_nb_states"])
        assert [w.shape[1] for w in wts] == list(g("counts")), row
        dm = np.abs(np.concatenate(means, 1) - g("means")).max()
        dsg = np.abs(np.concatenate([s[:, :, 0] for s in stds], 1) - g("stds")).max()
        dw = np.abs(np.concatenate(wts, 1) - g("logw")).max()
        assert dm < 1e-10 and dsg < 1e-10 and dw < 1e-9, (row, dm, dsg, dw)
        worst = max(worst, dm, dsg, dw)
    assert len(meta) == 40
    print("get_pos_PDF cases", len(meta), "worst difference", worst)
