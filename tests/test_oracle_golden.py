"""The numpy oracle (oracle/oracle_np.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import case_inputs
from oracle import oracle_np as O

TOL_LP = 1e-11     # abs, per sequence / per track log-probability
TOL_PRED = 1e-12   # abs, posteriors


def test_appendix_b_known_answers(appendix_b):
    b = appendix_b
    c = np.array(b["track"])[None]
    LE = np.array([[[b["LocErr"]]]])
    for row in b["rows"]:
        lpc = O.proba_cs(c, LE, b["ds"], b["Fs"], b["TrMat"], b["pBL"], row["isBL"], b["cell_dims"], row["ns"], row["F"], b["min_len"])
        assert abs(lpc[0] - row["LP_C"]) < TOL_LP
        if "preds0" in row:
            LP, preds = O.p_cs_inter_bound_stats(c, LE, b["ds"], b["Fs"], b["TrMat"], b["pBL"], row["isBL"], b["cell_dims"], 1, row["F"], 1, b["min_len"])
            assert LP.shape[1] == row["nB"]
            np.testing.assert_allclose(preds[0, :, 0], row["preds0"], atol=TOL_PRED, rtol=0)


def test_kernel_cases(kernel_cases):
    meta, data = kernel_cases
    assert len(meta) > 900
    worst_lp = worst_pr = 0.0
    for row in meta:
        if row["nB"] > 300000:
            continue  # covered by the C oracle test (memory/time)
        x = case_inputs(row, data)
        do_preds = 1 if row["ns"] == 1 else 0
        LP, preds = O.p_cs_inter_bound_stats(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], row["cell_dims"],
                                             row["ns"], row["F"], do_preds, row["min_len"])
        assert LP.shape[1] == row["nB"]
        if x["LP"] is not None:
            worst_lp = max(worst_lp, np.abs(LP - x["LP"]).max())
        mx = LP.max(1, keepdims=True)
        lpc = np.log(np.exp(LP - mx).sum(1)) + mx[:, 0]
        worst_lp = max(worst_lp, np.abs(lpc - x["LPC"]).max())
        if do_preds:
            worst_pr = max(worst_pr, np.abs(preds - x["preds"]).max())
    assert worst_lp < TOL_LP, worst_lp
    assert worst_pr < TOL_PRED, worst_pr


def _tracks(data, pre, keys):
    return {k: data[pre + k] for k in keys}


def test_end_to_end_objective(end_to_end):
    info, data = end_to_end
    e1 = info["e1"]
    tr = _tracks(data, "e1_tr_", e1["keys"])
    for name, ref in e1["cum"].items():
        F, ns = int(name[1]), int(name[-1])
        val = O.cum_proba_cs(e1["values"], tr, e1["dt"], e1["cell_dims"], None, ns, F, chunk=50)
        assert abs(val - ref) < 1e-9 * abs(ref), (name, val, ref)
    assert abs(e1["cum"]["F6_ns1"] - (-9476.422375154172)) < 1e-9  # SURVEY.md / BASELINE.md anchor
    lpc = O.cum_proba_cs(e1["values"], tr, e1["dt"], e1["cell_dims"], None, 1, 6, per_track=True)
    ref = np.concatenate([data["e1_lpc_F6_" + k] for k in e1["keys"]])
    np.testing.assert_allclose(lpc, ref, atol=TOL_LP, rtol=0)


def test_end_to_end_per_peak_locerr(end_to_end):
    info, data = end_to_end
    e2 = info["e2"]
    tr = _tracks(data, "e2_tr_", e2["keys"])
    sig = _tracks(data, "e2_sig_", e2["keys"])
    sl, of = e2["values"]["slope_LocErr"], e2["values"]["offset_LocErr"]
    aff = {k: np.clip(v * sl + of, 1e-6, np.inf) for k, v in sig.items()}  # tracking.py:928-930
    v = O.cum_proba_cs(e2["values_raw"], tr, e2["dt"], e2["cell_dims"], aff, 1, 4, chunk=50)
    assert abs(v - e2["cum_F4_ns1_affine"]) < 1e-9 * abs(v)
    v = O.cum_proba_cs(e2["values_raw"], tr, e2["dt"], e2["cell_dims"], sig, 1, 4, chunk=50)
    assert abs(v - e2["cum_F4_ns1_raw"]) < 1e-9 * abs(v)
    v = O.cum_proba_cs(e2["values_raw"], tr, e2["dt"], e2["cell_dims"], sig, 2, 3, chunk=50)
    assert abs(v - e2["cum_F3_ns2_raw"]) < 1e-9 * abs(v)


def test_end_to_end_predict(end_to_end):
    info, data = end_to_end
    e1, e2 = info["e1"], info["e2"]
    pr = O.predict_bs(e1["values"], _tracks(data, "e1_tr_", e1["keys"]), e1["dt"], e1["cell_dims"], 6)
    for k in e1["keys"]:
        np.testing.assert_allclose(pr[k], data["e1_pred_F6_" + k], atol=TOL_PRED, rtol=0)
    pr = O.predict_bs(e2["values_raw"], _tracks(data, "e2_tr_", e2["keys"]), e2["dt"], e2["cell_dims"], 4,
                      _tracks(data, "e2_sig_", e2["keys"]))
    for k in e2["keys"]:
        np.testing.assert_allclose(pr[k], data["e2_pred_F4_raw_" + k], atol=TOL_PRED, rtol=0)


def test_invalid_params_give_inf(end_to_end):
    info, data = end_to_end
    assert info["e3_inf"] == float("inf")
    bad = dict(D0=0.25, D1=1e-3, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    assert O.cum_proba_cs(bad, {"3": data["e1_tr_3"]}, 0.02, [1], None, 1, 4) == np.inf


def test_extract_params(params_plumbing):
    for row in params_plumbing["extract"]:
        if row["Matrix_type"] not in (0, 1):
            continue
        LocErr, ds, Fs, T, pBL = O.extract_params(row["values"], row["dt"], row["nb_substeps"], row["Matrix_type"])
        np.testing.assert_allclose(LocErr, row["LocErr"], rtol=0, atol=0)
        np.testing.assert_allclose(ds, row["ds"], rtol=1e-15)
        np.testing.assert_allclose(Fs, row["Fs"], rtol=0, atol=0)
        np.testing.assert_allclose(T, row["TrMat"], rtol=1e-15, atol=1e-17)
        assert pBL == row["pBL"]


def test_errors():
    with pytest.raises(ValueError):
        O.p_cs_inter_bound_stats(np.zeros((1, 1, 2)), np.array([[[0.02]]]), [0.01, 0.1], [0.5, 0.5], [[.9, .1], [.1, .9]])
    with pytest.raises(ValueError):
        O.p_cs_inter_bound_stats(np.zeros((1, 5, 2)), np.full((1, 3, 2), 0.02), [0.01, 0.1], [0.5, 0.5], [[.9, .1], [.1, .9]])


def test_c_oracle_kernel_cases(kernel_cases):
    """The plain-C restatement (oracle/extrack_oracle.c) against the same reference-generated vectors, including the cases too
    large for the vectorised numpy oracle (4 states, 3 substeps: 4^10 sequences per track)."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    from oracle import oracle_c
    meta, data = kernel_cases
    worst_lp = worst_pr = 0.0
    n = 0
    for row in meta:
        if row["id"] % 2 and row["nB"] < 300000:
            continue
        x = case_inputs(row, data)
        ps = O.p_stay_table(x["ds"], row["S"], row["ns"], row["cell_dims"])
        do_preds = row["ns"] == 1
        ll, pr = oracle_c.run(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], ps, row["ns"], row["F"], row["min_len"],
                              do_preds=do_preds, nthreads=4)
        worst_lp = max(worst_lp, np.abs(ll - x["LPC"]).max())
        if do_preds:
            worst_pr = max(worst_pr, np.abs(pr - x["preds"]).max())
        n += 1
    assert n > 450 and worst_lp < TOL_LP and worst_pr < TOL_PRED, (n, worst_lp, worst_pr)


def test_c_oracle_matches_numpy_oracle_on_affine_per_peak_locerr():
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    from oracle import oracle_c
    rng = np.random.default_rng(11)
    Cs = np.cumsum(rng.normal(0, 0.05, (7, 13, 2)), 1)
    sig = rng.uniform(0.01, 0.03, (7, 13, 2))
    ds, Fs, T = np.array([0.01, 0.08, 0.2]), np.array([.3, .3, .4]), np.array([[.8, .1, .1], [.1, .8, .1], [.1, .1, .8]])
    ps = O.p_stay_table(ds, 3, 1, [1.0])
    aff = np.clip(sig * 1.2 + 0.001, 1e-6, np.inf)
    ref = O.proba_cs(Cs, aff, ds, Fs, T, 0.05, 1, [1.0], 1, 4, 3)
    ll, _ = oracle_c.run(Cs, sig, ds, Fs, T, 0.05, 1, ps, 1, 4, 3, slope=1.2, offset=0.001)
    assert np.abs(ll - ref).max() < TOL_LP
