"""GPU parity tests proper: the HIP path, called through the C ABI, against golden vectors produced by the
reference and against the oracle on seeded inputs.  Tolerances (fp64): per-track LL abs 1e-10,
total rel 1e-12, posteriors abs 1e-9 (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import case_inputs

pytestmark = pytest.mark.gpu

TOL_LL = 1e-10
TOL_PRED = 1e-9


def _params(vals):
    from extrack_amd.lmfit_compat import Parameters
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


def test_library_loaded_and_device():
    from extrack_amd import _lib
    ctx = _lib.Context(0)
    assert ctx._lib.extrack_abi_version() == 6
    ctx.close()


def test_appendix_b(appendix_b):
    from extrack_amd import tracking as T
    b = appendix_b
    c = np.array(b["track"])[None]
    LE = np.array([[[b["LocErr"]]]])
    for row in b["rows"]:
        lpc = T.Proba_Cs(c, LE, b["ds"], b["Fs"], b["TrMat"], b["pBL"], row["isBL"], b["cell_dims"], row["ns"], row["F"], b["min_len"])
        assert abs(lpc[0] - row["LP_C"]) < TOL_LL
        if "preds0" in row:
            _, _, preds = T.P_Cs_inter_bound_stats(c, LE, b["ds"], b["Fs"], b["TrMat"], b["pBL"], row["isBL"], b["cell_dims"], 1,
                                                   row["F"], 1, b["min_len"])
            np.testing.assert_allclose(preds[0, :, 0], row["preds0"], atol=TOL_PRED, rtol=0)


def test_kernel_cases_golden(kernel_cases):
    """All 1004 reference-generated kernel cases: S in 2..5, ns in 1..3, D in 1..3, scalar / per-dim / per-peak
    localisation error, isBL 0/1, L from 2 (shorter than the window) to 30."""
    from extrack_amd import tracking as T
    meta, data = kernel_cases
    worst_ll = worst_pr = 0.0
    n = 0
    for row in meta:
        x = case_inputs(row, data)
        do_preds = row["ns"] == 1
        lp, _, preds = T.P_Cs_inter_bound_stats(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], row["cell_dims"],
                                                row["ns"], row["F"], int(do_preds), row["min_len"])
        d = np.abs(lp[:, 0] - x["LPC"]).max()
        assert d < TOL_LL, (row, d)
        worst_ll = max(worst_ll, d)
        if do_preds:
            dp = np.abs(preds - x["preds"]).max()
            assert dp < TOL_PRED, (row, dp)
            worst_pr = max(worst_pr, dp)
        n += 1
    print("cases", n, "worst |dLL|", worst_ll, "worst |dpred|", worst_pr)


def test_sequence_matrix_and_cur_Bs_golden(kernel_cases):
    """P_Cs_inter_bound_stats(return_matrix=True): the reference's full return contract (extrack/tracking.py:318) - LP[N, nB] against every
    matrix the reference produced for the golden cases (those with nB <= 256: 2-5 states, nb_substeps 1-3, isBL 0/1, per-dim and
    per-peak errors, tracks shorter and longer than the window), cur_Bs against the reference's digit layout (get_all_Bs, tracking.py:
    746-757) and the matrix's logsumexp against LP_C."""
    from extrack_amd import tracking as T
    meta, data = kernel_cases
    n, worst = 0, 0.0
    for row in meta:
        x = case_inputs(row, data)
        if x["LP"] is None:
            continue
        LP, cur_Bs, _ = T.P_Cs_inter_bound_stats(x["Cs"], x["LE"], x["ds"], x["Fs"], x["T"], row["pBL"], row["isBL"], row["cell_dims"],
                                                 row["ns"], row["F"], 0, row["min_len"], return_matrix=True)
        assert LP.shape == x["LP"].shape, (row, LP.shape)
        fin = np.isfinite(x["LP"]) & (x["LP"] > -650)
        worst = max(worst, np.abs(LP[fin] - x["LP"][fin]).max())
        assert np.all(LP[~fin] < -600)
        S, nd = row["S"], cur_Bs.shape[2]
        assert cur_Bs.shape == (1, LP.shape[1], nd) and S ** nd == LP.shape[1]
        i = np.arange(LP.shape[1])
        assert all(np.array_equal(cur_Bs[0, :, k], (i // S ** k) % S) for k in range(nd))
        mx = LP.max(1, keepdims=True)
        assert np.abs(np.log(np.exp(LP - mx).sum(1)) + mx[:, 0] - x["LPC"]).max() < TOL_LL
        n += 1
    print("sequence matrices checked:", n, "worst |dLP|", worst)
    assert n >= 200 and worst < 1e-9, (n, worst)


def _tracks(data, pre, keys):
    return {k: data[pre + k] for k in keys}


def test_end_to_end_objective_and_predict(end_to_end):
    from extrack_amd import tracking as T
    info, data = end_to_end
    e1 = info["e1"]
    tr = _tracks(data, "e1_tr_", e1["keys"])
    p = _params(e1["values"])
    _, lst, _ = T.engine.sort_buckets(tr)
    for name, ref in e1["cum"].items():
        F, ns = int(name[1]), int(name[-1])
        val = T.cum_Proba_Cs(p, lst, e1["dt"], e1["cell_dims"], None, 2, ns, F, verbose=0)
        assert abs(val - ref) < 1e-12 * abs(ref), (name, val, ref)
    pr = T.predict_Bs(tr, e1["dt"], p, cell_dims=e1["cell_dims"], nb_states=2, frame_len=6)
    for k in e1["keys"]:
        np.testing.assert_allclose(pr[k], data["e1_pred_F6_" + k], atol=TOL_PRED, rtol=0)


def test_end_to_end_per_peak_locerr(end_to_end):
    from extrack_amd import tracking as T
    info, data = end_to_end
    e2 = info["e2"]
    tr = _tracks(data, "e2_tr_", e2["keys"])
    sig = _tracks(data, "e2_sig_", e2["keys"])
    _, lst, lsig = T.engine.sort_buckets(tr, sig)
    v = T.cum_Proba_Cs(_params(e2["values"]), lst, e2["dt"], e2["cell_dims"], lsig, 3, 1, 4, verbose=0)
    assert abs(v - e2["cum_F4_ns1_affine"]) < 1e-12 * abs(v)
    praw = _params(e2["values_raw"])
    v = T.cum_Proba_Cs(praw, lst, e2["dt"], e2["cell_dims"], lsig, 3, 1, 4, verbose=0)
    assert abs(v - e2["cum_F4_ns1_raw"]) < 1e-12 * abs(v)
    v = T.cum_Proba_Cs(praw, lst, e2["dt"], e2["cell_dims"], lsig, 3, 2, 3, verbose=0)
    assert abs(v - e2["cum_F3_ns2_raw"]) < 1e-12 * abs(v)
    pr = T.predict_Bs(tr, e2["dt"], praw, cell_dims=e2["cell_dims"], nb_states=3, frame_len=4, input_LocErr=sig)
    for k in e2["keys"]:
        np.testing.assert_allclose(pr[k], data["e2_pred_F4_raw_" + k], atol=TOL_PRED, rtol=0)


def test_invalid_params_inf(end_to_end):
    from extrack_amd import tracking as T
    _, data = end_to_end
    bad = _params(dict(D0=0.25, D1=1e-3, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1))
    assert T.cum_Proba_Cs(bad, [data["e1_tr_3"]], 0.02, [1], None, 2, 1, 4, verbose=0) == np.inf


def test_c1_config_objective():
    """BASELINE.json configs[0]: sim_FOV(10k) fixture, reference objective value."""
    import json
    import os
    from conftest import GOLDEN
    from extrack_amd import tracking as T
    info = json.load(open(os.path.join(GOLDEN, "c1_simfov_10k.json")))
    data = np.load(os.path.join(GOLDEN, "c1_simfov_10k.npz"))
    tr = {k: data["tr_" + k] for k in info["keys"]}
    _, lst, _ = T.engine.sort_buckets(tr)
    v = T.cum_Proba_Cs(_params(info["values"]), lst, info["dt"], info["cell_dims"], None, 2, 1, info["frame_len"], verbose=0)
    assert abs(v - info["cum_Proba_Cs"]) < 1e-12 * abs(v), (v, info["cum_Proba_Cs"])


def test_c1_param_fitting_recovers_simulated_parameters(capsys):
    """BASELINE.json configs[0] end to end: sim_FOV(10k) fixture -> param_fitting (own lmfit-compatible BFGS) on the GPU.
    The reference validates the same way (Tutorial_ExTrack.ipynb:788-792: D1 0.2517 vs 0.25, LocErr 0.01998 vs 0.02,
    p01 0.0951 vs 0.1); simulated truth: D=[0, 0.25], LocErr 0.02, F0 0.6, p01 = p10 = 0.1, pBL 0.1."""
    import json
    import os
    from conftest import GOLDEN
    from extrack_amd import tracking as T
    info = json.load(open(os.path.join(GOLDEN, "c1_simfov_10k.json")))
    data = np.load(os.path.join(GOLDEN, "c1_simfov_10k.npz"))
    tr = {k: data["tr_" + k] for k in info["keys"]}
    p0 = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1],
                           estimated_Fs=[0.5], estimated_transition_rates=0.05)
    fit = T.param_fitting(tr, info["dt"], params=p0, nb_states=2, nb_substeps=1, frame_len=6, verbose=0, method="bfgs", cell_dims=[1])
    capsys.readouterr()
    v = {k: fit.params[k].value for k in fit.params}
    assert fit.residual.shape == (1,)
    assert fit.residual[0] <= info["cum_Proba_Cs"] + 1e-6  # at least as good as the generating parameters
    assert abs(v["D1"] - 0.25) < 0.02 and v["D0"] < 5e-3
    assert abs(v["LocErr"] - 0.02) < 0.002
    assert abs(v["F0"] - 0.6) < 0.05 and abs(v["F1"] - (1 - v["F0"])) < 1e-12
    assert abs(v["p01"] - 0.1) < 0.03 and abs(v["p10"] - 0.1) < 0.03
    assert 10 < fit.nfev < 5000  # ~30 evaluations with the analytic gradient (default), ~280 when differenced numerically


@pytest.mark.parametrize("S,ns,F,L,N", [(2, 1, 6, 30, 5000), (3, 1, 6, 17, 600), (4, 1, 5, 20, 200), (4, 3, 4, 12, 24), (2, 2, 6, 40, 500),
                                         (3, 1, 4, 50, 700), (2, 1, 9, 25, 300), (4, 1, 6, 14, 40),
                                         (5, 1, 6, 8, 8), (6, 1, 5, 7, 6), (2, 1, 12, 16, 70), (7, 1, 3, 8, 12), (8, 1, 3, 6, 9), (3, 2, 7, 9, 5)])
def test_seeded_vs_oracle(S, ns, F, L, N):
    """Seeded synthetic batches at sizes the numpy oracle finishes in seconds (incl. multi-wave tracks:
    S=3,F=6 -> 243 groups; S=4,F=6 -> 1024 groups; S=2,F=9 -> 256 groups).  Round 4: the models the LDS kernels refuse - 5 states at the
    reference's default frame_len 6 (15 625 sequences per track), 6 states at frame_len 5, 2 states at frame_len 12 (2048 groups), posteriors
    with 7 / 8 states, 3 states with 2 substeps at frame_len 7 - run through the global-state kernel (csrc/xt_big.h), likelihood and
    posteriors (reference: extrack/tracking.py:109-318, which has no size limit)."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    rng = np.random.default_rng(S * 100 + F)
    Ds = np.sort(rng.uniform(0.0, 0.3, S))
    Ds[0] = 0.0
    Tm = np.full((S, S), 0.05)
    Tm[np.arange(S), np.arange(S)] = 1 - 0.05 * (S - 1)
    Fs = np.full(S, 1.0 / S)
    Cs = synth.brownian_tracks(N, L, Ds, Tm, Fs, seed=S + F)
    ds = np.sqrt(2 * Ds * 0.02) + 1e-3
    LE = np.array([[[0.02]]])
    TT = Tm.copy()
    ref = np.concatenate([O.proba_cs(Cs[a:a + 100], LE, ds, Fs, TT, 0.1, 1, [1.0], ns, F, 3) for a in range(0, N, 100)])
    got = T.Proba_Cs(Cs, LE, ds, Fs, TT, 0.1, 1, [1.0], ns, F, 3)
    assert np.abs(got - ref).max() < TOL_LL, np.abs(got - ref).max()
    if ns == 1 and N <= 1000:
        _, _, preds = T.P_Cs_inter_bound_stats(Cs[:100], LE, ds, Fs, TT, 0.1, 0, [1.0], 1, F, 1, 3)
        _, pref = O.p_cs_inter_bound_stats(Cs[:100], LE, ds, Fs, TT, 0.1, 0, [1.0], 1, F, 1, 3)
        assert np.abs(preds - pref).max() < TOL_PRED


def test_global_state_kernel_forced_on_ordinary_models(monkeypatch):
    """csrc/xt_big.h forced (EXTRACK_FORCE_BIG=1) on models the LDS kernels serve: same per-track LL (1e-10) and posteriors (1e-9) as those
    kernels' oracle, several wavefronts and a ragged last batch, two length buckets in one launch."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    monkeypatch.setenv("EXTRACK_FORCE_BIG", "1")
    vals = dict(D0=1e-3, D1=0.05, D2=0.3, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.08, p02=0.04, p10=0.06, p12=0.05, p20=0.03, p21=0.07, pBL=0.1)
    Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    tracks = {"9": synth.brownian_tracks(333, 9, [0.0, 0.05, 0.3], Tm, [0.3, 0.3, 0.4], seed=1),
              "14": synth.brownian_tracks(150, 14, [0.0, 0.05, 0.3], Tm, [0.3, 0.3, 0.4], seed=2)}
    p = _params(vals)
    _, lst, _ = T.engine.sort_buckets(tracks)
    got = T.cum_Proba_Cs(p, lst, 0.02, [1], None, 3, 1, 4, verbose=0)
    ref = O.cum_proba_cs(vals, tracks, 0.02, [1], None, 1, 4)
    assert abs(got - ref) < 1e-12 * abs(ref), (got, ref)
    pr = T.predict_Bs(tracks, 0.02, p, cell_dims=[1], nb_states=3, frame_len=4)
    pro = O.predict_bs(vals, tracks, 0.02, [1], 4)
    assert max(np.abs(pr[k] - pro[k]).max() for k in tracks) < 1e-9


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (1e6 x 30): size-independent properties.
    (a) total == sum of per-track values (rel 1e-12); (b) permutation invariance of the total;
    (c) the total of a 5e5 + 5e5 split equals the whole (additivity over shards, what the multi-GPU path relies on);
    (d) a 2000-track slice matches the oracle."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    N, L = 1000000, 30
    Cs = synth.brownian_tracks(N, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
    vals = dict(D0=0.0, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    p = _params(vals)
    ts = T.TrackSet([Cs])
    model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
    tot, per = ts.loglik(model, per_track=True)
    ts.close()
    assert np.all(np.isfinite(per))
    assert abs(tot - per.sum()) < 1e-12 * abs(tot)
    ref = O.proba_cs(Cs[:2000], np.array([[[0.02]]]), np.sqrt(2 * np.array([0.0, 0.25]) * 0.02), [.6, .4],
                     T.extract_params(p, 0.02, 2, 1)[3], 0.1, 0, [1], 1, 6, 30)
    assert np.abs(per[:2000] - ref).max() < TOL_LL
    perm = np.random.default_rng(1).permutation(N)
    ts2 = T.TrackSet([Cs[perm]])
    tot2 = ts2.loglik(T._objective_model(p, ts2, 0.02, [1], None, 2, 1, 6, 1))
    ts2.close()
    assert abs(tot2 - tot) < 1e-12 * abs(tot)
    parts = 0.0
    for a, b in ((0, N // 2), (N // 2, N)):
        t3 = T.TrackSet([Cs[a:b]])
        parts += t3.loglik(T._objective_model(p, t3, 0.02, [1], None, 2, 1, 6, 1))
        t3.close()
    assert abs(parts - tot) < 1e-12 * abs(tot)


def test_rccl_single_rank_allreduce_path():
    """The multi-GPU code path (kernels and the RCCL all-reduce enqueued on one stream, scalar left in device memory)
    exercised with a 1-rank nccl group - all this box has.  Runs in a subprocess with a hard timeout: an RCCL bootstrap
    that stalls on the host's network configuration must not hang the suite (reported as a FAILURE)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", GLOO_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_check.py")], env=env, capture_output=True,
                           text=True, timeout=240)
    except subprocess.TimeoutExpired:
        # the only nccl-backend coverage of the suite: a stall is a failure to look at, not a skip (set EXTRACK_ALLOW_RCCL_SKIP=1 on a host
        # whose network configuration is known to stall the bootstrap)
        if os.environ.get("EXTRACK_ALLOW_RCCL_SKIP") == "1":
            pytest.skip("RCCL bootstrap did not complete within 240 s on this host")
        pytest.fail("RCCL bootstrap / single-rank communicator check did not complete within 240 s")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RCCL_PATH_OK" in r.stdout


def _oracle_total(vals, tracks, dt, cell_dims, F, ns=1, sig=None):
    from oracle import oracle_np as O
    return O.cum_proba_cs(vals, tracks, dt, cell_dims, sig, ns, F)


def test_input_layouts_and_dtypes_give_identical_results():
    """float32 / Fortran-ordered / strided views of the same numbers must give the result of the contiguous fp64 copy."""
    from extrack_amd import synth, tracking as T
    base = synth.brownian_tracks(500, 12, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=3).astype(np.float32).astype(np.float64)
    p = _params(dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1))
    ref = T.cum_Proba_Cs(p, [base], 0.02, [1], None, 2, 1, 6, verbose=0)
    big = np.zeros((500, 24, 2))
    big[:, ::2] = base
    for variant in (base.astype(np.float32), np.asfortranarray(base), big[:, ::2]):
        _, lst, _ = T.engine.sort_buckets({"12": variant})
        assert T.cum_Proba_Cs(p, lst, 0.02, [1], None, 2, 1, 6, verbose=0) == ref


def test_nan_input_maps_to_inf_like_the_reference(capsys):
    from extrack_amd import synth, tracking as T
    Cs = synth.brownian_tracks(64, 9, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=4)
    Cs[17, 3, 1] = np.nan
    p = _params(dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1))
    assert T.cum_Proba_Cs(p, [Cs], 0.02, [1], None, 2, 1, 6, verbose=0) == np.inf  # tracking.py:1084-1086
    capsys.readouterr()


@pytest.mark.parametrize("S,F", [(2, 6), (3, 4)])
def test_extreme_displacements_do_not_underflow(S, F):
    """A 40 um jump with 20 nm localisation error has a Gaussian exponent of about -1e6: exp() underflows in a naive linear-domain
    implementation; the extended-range weights must reproduce the reference's (finite) log-domain value."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    rng = np.random.default_rng(9)
    Ds = [0.0, 0.25, 1.0][:S]
    Tm = np.full((S, S), 0.1)
    Tm[np.arange(S), np.arange(S)] = 1 - 0.1 * (S - 1)
    Fs = np.full(S, 1.0 / S)
    Cs = synth.brownian_tracks(40, 15, Ds, Tm, Fs, seed=5)
    Cs[::4, 7:] += 40.0                      # one huge jump in a quarter of the tracks
    Cs[1::4, 3] += rng.normal(0, 5.0, (10, 2))  # an outlier position
    ds = np.sqrt(2 * np.array(Ds) * 0.02) + 1e-4
    LE = np.array([[[0.02]]])
    ref = O.proba_cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 3)
    got = T.Proba_Cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 3)
    assert np.all(np.isfinite(got)) and ref.min() < -1e4
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-10)


@pytest.mark.parametrize("S,F", [(2, 6), (3, 4)])
def test_absurd_jump_clamps_instead_of_wrapping(S, F):
    """A jump of tens of thousands of localisation errors (Gaussian exponent below -1.1e7, the clamp of the table-driven exp): the
    weight must become (practically) zero, not wrap around the int32 exponent.  Other tracks of the launch are unaffected."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    Ds = [0.0, 0.25, 1.0][:S]
    Tm = np.full((S, S), 0.1)
    Tm[np.arange(S), np.arange(S)] = 1 - 0.1 * (S - 1)
    Fs = np.full(S, 1.0 / S)
    Cs = synth.brownian_tracks(12, 15, Ds, Tm, Fs, seed=6)
    Cs[3, 7:] += 3000.0
    ds = np.sqrt(2 * np.array(Ds) * 0.02) + 1e-4
    LE = np.array([[[0.02]]])
    ref = O.proba_cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 3)
    got = T.Proba_Cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 3)
    ok = np.arange(12) != 3
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-13, atol=1e-10)
    assert np.isfinite(got[3]) and got[3] < -7e6 and ref[3] < -7e6, (got[3], ref[3])


def test_tiny_and_zero_transition_probabilities():
    """Transition probabilities far below the 'well-scaled' bounds of the 2-state fast path (1e-200, 1e-300: the guarded steps normalise
    every stored weight) against the oracle, and an exactly zero one (absorbing state: the reference's log-domain code returns NaN and its
    objective refuses such models) against the 1e-300 result - the extra sequences contribute nothing at double precision."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    Fs = np.array([0.3, 0.7])
    Cs = synth.brownian_tracks(50, 14, [0.0, 0.25], [[0.9, 0.1], [0.15, 0.85]], Fs, seed=3)
    ds = np.sqrt(2 * np.array([1e-4, 0.25]) * 0.02)
    LE = np.array([[[0.02]]])
    out = {}
    for eps in (0.0, 1e-300, 1e-200, 1e-25):
        Tm = np.array([[1.0 - eps, eps], [0.15, 0.85]])
        out[eps] = T.Proba_Cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, 6, 3)
        if eps > 0:
            ref = O.proba_cs(Cs, LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, 6, 3)
            np.testing.assert_allclose(out[eps], ref, rtol=1e-13, atol=1e-10)
    assert np.all(np.isfinite(out[0.0]))
    np.testing.assert_allclose(out[0.0], out[1e-300], rtol=1e-13, atol=1e-10)


def test_large_coordinate_offsets():
    """Positions around 1e4 um (pixel-like coordinates): both implementations difference nearby fp64 numbers; parity must hold to the
    conditioning of the problem (|c| * eps / sigma^2 ~ 1e-12 * 1e4 / 4e-4 per step)."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    Cs = synth.brownian_tracks(200, 20, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=6) + 1.0e4
    ds = np.sqrt(2 * np.array([0.0, 0.25]) * 0.02) + 1e-4
    LE = np.array([[[0.02]]])
    Tm = np.array([[.9, .1], [.1, .9]])
    ref = O.proba_cs(Cs, LE, ds, [.6, .4], Tm, 0.1, 1, [1.0], 1, 6, 3)
    got = T.Proba_Cs(Cs, LE, ds, [.6, .4], Tm, 0.1, 1, [1.0], 1, 6, 3)
    assert np.abs(got - ref).max() < 1e-6


def test_per_dimension_locerr_and_3d_through_public_api():
    from extrack_amd import synth, tracking as T
    tr = {str(L): synth.brownian_tracks(150, L, [0.0, 0.1, 0.4], [[.9, .05, .05], [.05, .9, .05], [.05, .05, .9]], [.3, .3, .4], dims=3, seed=L)
          for L in (4, 8, 13)}
    vals = dict(D0=1e-4, D1=0.1, D2=0.4, LocErr0=0.02, LocErr1=0.02, LocErr2=0.05, F0=0.3, F1=0.3, F2=0.4, p01=0.05, p02=0.05, p10=0.05,
                p12=0.05, p20=0.05, p21=0.05, pBL=0.1)
    p = _params(vals)
    _, lst, _ = T.engine.sort_buckets(tr)
    for ns, F in ((1, 4), (2, 3)):
        got = T.cum_Proba_Cs(p, lst, 0.03, [1.0, 2.0], None, 3, ns, F, verbose=0)
        ref = _oracle_total(vals, tr, 0.03, [1.0, 2.0], F, ns)
        assert abs(got - ref) < 1e-12 * abs(ref)
    pr = T.predict_Bs(tr, 0.03, p, cell_dims=[1.0, 2.0], nb_states=3, frame_len=4)
    from oracle import oracle_np as O
    pro = O.predict_bs(vals, tr, 0.03, [1.0, 2.0], 4)
    for k in tr:
        assert pr[k].shape == (150, int(k), 3)
        np.testing.assert_allclose(pr[k], pro[k], atol=TOL_PRED, rtol=0)


def test_tiny_and_ragged_datasets():
    """One track; buckets smaller than a wave; a bucket with a single 2-position track; empty bucket keys preserved by predict_Bs."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    p = _params(vals)
    tr = {"2": synth.brownian_tracks(1, 2, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1),
          "3": np.zeros((0, 3, 2)),
          "7": synth.brownian_tracks(3, 7, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=2),
          "31": synth.brownian_tracks(65, 31, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=3)}
    _, lst, _ = T.engine.sort_buckets(tr)
    got = T.cum_Proba_Cs(p, lst, 0.02, [1], None, 2, 1, 6, verbose=0)
    ref = O.cum_proba_cs(vals, tr, 0.02, [1], None, 1, 6)
    assert abs(got - ref) < 1e-12 * abs(ref)
    pr = T.predict_Bs(tr, 0.02, p, cell_dims=[1], nb_states=2, frame_len=5)
    pro = O.predict_bs(vals, tr, 0.02, [1], 5)
    assert sorted(pr.keys(), key=int) == ["2", "3", "7", "31"] and pr["3"].shape == (0, 3, 2)
    for k in ("2", "7", "31"):
        np.testing.assert_allclose(pr[k], pro[k], atol=TOL_PRED, rtol=0)


@pytest.mark.parametrize("S,F,L", [(2, 6, 1000), (2, 4, 257), (3, 4, 400), (4, 3, 300)])
def test_long_tracks_error_accumulation(S, F, L):
    """Rounding errors accumulate along the track: parity on tracks of hundreds of positions (many 32-position staging chunks)."""
    from extrack_amd import synth, tracking as T
    from oracle import oracle_np as O
    Ds = [0.0, 0.05, 0.25, 0.8][:S]
    Tm = np.full((S, S), 0.06)
    Tm[np.arange(S), np.arange(S)] = 1 - 0.06 * (S - 1)
    Fs = np.full(S, 1.0 / S)
    Cs = synth.brownian_tracks(24, L, Ds, Tm, Fs, seed=L)
    ds = np.sqrt(2 * np.array(Ds) * 0.02) + 1e-4
    LE = np.array([[[0.02]]])
    ref = O.proba_cs(Cs, LE, ds, Fs, Tm, 0.1, 0, [1.0], 1, F, 3)
    got = T.Proba_Cs(Cs, LE, ds, Fs, Tm, 0.1, 0, [1.0], 1, F, 3)
    assert np.abs(got - ref).max() < TOL_LL, np.abs(got - ref).max()
    _, _, preds = T.P_Cs_inter_bound_stats(Cs[:6], LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 1, 3)
    _, pref = O.p_cs_inter_bound_stats(Cs[:6], LE, ds, Fs, Tm, 0.1, 1, [1.0], 1, F, 1, 3)
    assert np.abs(preds - pref).max() < TOL_PRED
