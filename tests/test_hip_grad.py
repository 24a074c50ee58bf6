"""GPU tests of the fused log-likelihood + gradient kernel (extrack_loglik_grad): the gradient through the
C ABI against Richardson-extrapolated central differences of the pinned oracle (<= 1e-6 relative on every direction), the
parameter-level gradient of the fit objective, and fits that reach the optimum of the finite-difference path with several times
fewer objective calls."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, case_inputs
from test_grad_cpu import model_directions, oracle_fd_gradient

pytestmark = pytest.mark.gpu


def _params(vals):
    from extrack_amd.lmfit_compat import Parameters
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


@pytest.mark.parametrize("gradr", [None, "4", "3", "rev"])
def test_gradient_vs_oracle_central_differences_on_golden_models(kernel_cases, gradr, monkeypatch):
    """gradr: None = the launcher's choice of kernel per model (xt_reg2.h for 2 states with a global error, xt_gradr.h up to 4 members per group and 256 groups per track, else xt_grad.h);
    "4" / "3" = the register-resident general kernels (xt_gradr.h) forced for every model they serve, 4 / 3 directions per pass.
    Golden kernel cases (reference-generated inputs: 2-4 states, nb_substeps 1-2, 1-3 dims, scalar / per-dim localisation error,
    isBL 0/1, tracks shorter and longer than the window): LL must equal the golden LP_C, the gradient along EVERY model direction
    must match central differences of the oracle to 1e-6 relative."""
    from extrack_amd import tracking as T
    from oracle import oracle_np as O
    if gradr == "rev":
        monkeypatch.setenv("EXTRACK_GRAD_PATH", "rev")  # the reverse-mode kernels (xt_rev.h) for every model they serve (default: 3 / 4 members per group)
    elif gradr:
        monkeypatch.setenv("EXTRACK_GRADR_NPC", gradr)  # read when a context is created
        monkeypatch.setenv("EXTRACK_GRAD_PATH", "gradr")
    meta, data = kernel_cases
    done, seen = 0, set()
    for row in meta:
        x = case_inputs(row, data)
        Cs, LE = x["Cs"], x["LE"]
        S, ns, F, D = len(x["ds"]), row["ns"], row["F"], Cs.shape[2]
        key = (S, ns, F, D, LE.shape[2], row["isBL"], Cs.shape[1] > F)
        if LE.shape[1] != 1 or S > 4 or S ** F > 300 or key in seen or Cs.shape[1] < 3 or np.any(x["ds"] <= 0):
            continue
        seen.add(key)
        K = LE.shape[2]
        le, ds2, cell = LE[0, 0].astype(float), np.asarray(x["ds"], float) ** 2, row["cell_dims"]
        dirs = model_directions(S, K, ns, ds2, x["T"], le, cell)
        ts, _ = T._one_bucket(Cs, LE, row["isBL"], row["min_len"], 0)
        try:
            model = ts.make_model(LE, x["ds"], x["Fs"], x["T"], row["pBL"], cell, ns, F)
            ll, g = ts.ctx.loglik_grad(model, [d[1] for d in dirs])
        finally:
            ts.close()
        assert abs(ll - x["LPC"].sum()) < 1e-10 * max(1.0, abs(ll))
        fd = oracle_fd_gradient(Cs, le, ds2, np.asarray(x["Fs"], float), np.asarray(x["T"], float), row["pBL"], row["isBL"], cell, ns, F,
                                row["min_len"], dirs)
        rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
        assert rel.max() < 1e-6, (row, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6])
        done += 1
        if done >= 40:
            break
    assert done >= 20, done
    print("gradient checked on", done, "golden models")


def test_parameter_gradient_of_the_fit_objective_c1():
    """configs[0] fixture (sim_FOV 10k, 16 length buckets): d(-sum LL)/d(free parameter values) from ONE gradient evaluation against
    central differences of the oracle's cum_proba_cs through extract_params (rates -> probabilities, F1 = 1 - F0, D -> ds -> p_stay)."""
    from extrack_amd import tracking as T
    from oracle import oracle_np as O
    info = json.load(open(os.path.join(GOLDEN, "c1_simfov_10k.json")))
    data = np.load(os.path.join(GOLDEN, "c1_simfov_10k.npz"))
    tr = {k: data["tr_" + k][:300] for k in info["keys"]}
    p = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.002, 0.2], estimated_Fs=[0.55],
                          estimated_transition_rates=[0.08, 0.12])
    p["pBL"].value = 0.07
    _, lst, _ = T.engine.sort_buckets(tr)
    names = [k for k in p if p[k].vary and p[k].expr is None]
    val, g = T.cum_Proba_Cs_grad(p, names, lst, info["dt"], info["cell_dims"], None, 2, 1, 6, verbose=0)
    vals = {k: p[k].value for k in p}
    assert abs(val - O.cum_proba_cs(vals, tr, info["dt"], info["cell_dims"], None, 1, 6)) < 1e-12 * abs(val)
    for n, gi in zip(names, g):
        h = 1e-4 * max(abs(vals[n]), 1e-3)

        def f(x):
            v = dict(vals)
            v[n] = vals[n] + x
            v["F1"] = 1 - v["F0"]
            return O.cum_proba_cs(v, tr, info["dt"], info["cell_dims"], None, 1, 6)

        fd = (4 * (f(h / 2) - f(-h / 2)) / h - (f(h) - f(-h)) / (2 * h)) / 3
        assert abs(gi - fd) < 1e-6 * max(abs(fd), 1e-3 * np.abs(g).max()), (n, gi, fd)


@pytest.mark.parametrize("path", [None, "gradr", "lds"])
@pytest.mark.parametrize("S", [2, 3])
def test_parameter_gradient_with_per_peak_errors(S, path, monkeypatch):
    """input_LocErr + LocErr_type 4 (error = clip(sigma * slope + offset), extrack/tracking.py:946-955): d(-sum LL)/d(free parameters) incl.
    slope and offset from one gradient evaluation - path None: the launcher's choice (reverse mode xt_rev.h), "gradr" / "lds": the
    forward-mode kernel families - against Richardson central differences of the SAME context's objective (whose values the parity
    tests pin to the oracle)."""
    from extrack_amd import gradient, synth, tracking as T
    if path:
        monkeypatch.setenv("EXTRACK_GRAD_PATH", path)
    rng = np.random.default_rng(S)
    Ds = [0.0, 0.05, 0.3][:S] if S == 3 else [0.0, 0.25]
    Tm = np.full((S, S), 0.06) + np.eye(S) * (1 - 0.06 * S)
    tr, sg = {}, {}
    for L in (6, 9, 14):
        tr[str(L)] = synth.brownian_tracks(40, L, Ds, Tm, [1.0 / S] * S, seed=L)
        sg[str(L)] = rng.uniform(0.012, 0.03, (40, L, 1))
    p = T.generate_params(nb_states=S, LocErr_type=4, estimated_Ds=[1e-4, 0.05, 0.3][:S] if S == 3 else [1e-4, 0.25], estimated_Fs=[1.0 / S] * (S - 1),
                          estimated_transition_rates=0.07, slope_offsets_estimates=[1.1, 0.002])
    names = gradient.free_names(p)
    assert "slope_LocErr" in names and "offset_LocErr" in names
    _, lst, sig = T.engine.sort_buckets(tr, sg)
    ts = T.TrackSet(lst, sig)
    try:
        v, g = gradient.objective_and_gradient(p, ts, 0.02, [1.0], S, 1, 5, names=names)

        def f(n, x):
            q = p.copy()
            q[n].value = p[n].value + x
            q.update_constraints()
            return -ts.loglik(T._objective_model(q, ts, 0.02, [1.0], sig, S, 1, 5, 1))

        assert abs(v - f(names[0], 0.0)) < 1e-12 * abs(v)
        for n, gi in zip(names, g):
            h = 1e-3 * max(abs(p[n].value), 1e-3)
            fd = (4 * (f(n, h / 2) - f(n, -h / 2)) / h - (f(n, h) - f(n, -h)) / (2 * h)) / 3
            assert abs(gi - fd) < 2e-6 * max(abs(fd), 1e-3 * np.abs(g).max()), (n, gi, fd)
    finally:
        ts.close()


def test_reverse_mode_log_budget_falls_back_to_forward_mode(monkeypatch):
    """The reverse-mode kernels log the merged state of every step per track slot (device memory proportional to the track length); when
    the budget (EXTRACK_REV_LOG_MB) does not hold one block per two CUs the launcher takes the forward-mode kernels: same gradient."""
    from extrack_amd import gradient, synth, tracking as T
    Tm = np.full((3, 3), 0.05) + np.eye(3) * 0.85
    tr = {"200": synth.brownian_tracks(64, 200, [0.0, 0.05, 0.3], Tm, [0.3, 0.3, 0.4], seed=4)}
    p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.05, 0.3], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
    names = gradient.free_names(p)
    _, lst, _ = T.engine.sort_buckets(tr)
    out = {}
    for mb in ("16384", "1"):
        monkeypatch.setenv("EXTRACK_REV_LOG_MB", mb)
        ts = T.TrackSet(lst)
        try:
            out[mb] = gradient.objective_and_gradient(p, ts, 0.02, [1.0], 3, 1, 5, names=names)
            info = ts.ctx.last_launch_info()
        finally:
            ts.close()
        out[mb + "lds"] = info["lds_bytes"]
    assert out["16384lds"] != out["1lds"]  # two different kernel families served the two calls
    assert abs(out["1"][0] - out["16384"][0]) < 1e-12 * abs(out["1"][0])
    assert np.abs(out["1"][1] - out["16384"][1]).max() < 1e-9 * np.abs(out["1"][1]).max()


def test_c1_fit_analytic_gradient_same_optimum_fewer_calls(capsys):
    """configs[0] end to end: param_fitting with the analytic gradient vs the reference-style finite-difference BFGS on the same
    data and starting point: same optimum (objective within 1e-6 relative, parameters within 1 %), >= 5x fewer objective calls."""
    from extrack_amd import tracking as T
    info = json.load(open(os.path.join(GOLDEN, "c1_simfov_10k.json")))
    data = np.load(os.path.join(GOLDEN, "c1_simfov_10k.npz"))
    tr = {k: data["tr_" + k] for k in info["keys"]}
    p0 = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1],
                           estimated_Fs=[0.5], estimated_transition_rates=0.05)
    fa = T.param_fitting(tr, info["dt"], params=p0, nb_states=2, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="analytic")
    ff = T.param_fitting(tr, info["dt"], params=p0, nb_states=2, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
    capsys.readouterr()
    print("C1 fit: analytic nfev %d (residual %.6f), fd nfev %d (residual %.6f)" % (fa.nfev, fa.residual[0], ff.nfev, ff.residual[0]))
    assert fa.residual[0] <= ff.residual[0] + 1e-6 * abs(ff.residual[0])
    for k in ("D1", "LocErr", "F0", "p01", "p10"):
        assert abs(fa.params[k].value - ff.params[k].value) < 0.01 * abs(ff.params[k].value) + 1e-5, k
    assert fa.nfev * 5 <= ff.nfev, (fa.nfev, ff.nfev)
    assert abs(fa.params["D1"].value - 0.25) < 0.02 and abs(fa.params["LocErr"].value - 0.02) < 0.002


def test_c3_fit_analytic_gradient_same_optimum_fewer_calls(capsys):
    """Reduced-size configs[2] (3 states, 46 buckets of lengths 5-50, 6e4 tracks, 13 free parameters, the reference's default
    frame_len = 6): analytic-gradient fit vs finite-difference fit."""
    from extrack_amd import tracking as T
    from test_hip_configs import _c3_tracks
    tracks = _c3_tracks(6e4, seed0=500)
    p0 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                           estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.1)
    fa = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="analytic")
    ff = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
    capsys.readouterr()
    print("C3 fit: analytic nfev %d (residual %.6f), fd nfev %d (residual %.6f)" % (fa.nfev, fa.residual[0], ff.nfev, ff.residual[0]))
    assert fa.residual[0] <= ff.residual[0] + 1e-6 * abs(ff.residual[0])
    assert fa.nfev * 5 <= ff.nfev, (fa.nfev, ff.nfev)
    assert abs(fa.params["D1"].value - 0.04) < 0.006 and abs(fa.params["D2"].value - 0.25) < 0.02


def test_gradient_full_size_properties():
    """configs[1] at full size (1e6 x 30): the LL of the gradient kernel equals extrack_loglik's, the gradient is additive over two
    row shards and reproducible bit for bit between two calls."""
    from extrack_amd import gradient, synth, tracking as T
    N, L = 1000000, 30
    Cs = synth.brownian_tracks(N, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
    p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[0.001, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6],
                          estimated_transition_rates=0.1)
    names = gradient.free_names(p)
    ts = T.TrackSet([Cs])
    v1, g1 = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, 6, names=names)
    v2, g2 = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, 6, names=names)
    ll = ts.loglik(T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1))
    ms = ts.ctx.last_grad_ms()
    ts.close()
    assert v1 == v2 and np.array_equal(g1, g2) and np.all(np.isfinite(g1))
    assert abs(v1 + ll) < 1e-12 * abs(ll)
    parts_v, parts_g = 0.0, 0.0
    for a, b in ((0, N // 2), (N // 2, N)):
        t3 = T.TrackSet([Cs[a:b]])
        v, g = gradient.objective_and_gradient(p, t3, 0.02, [1], 2, 1, 6, names=names)
        t3.close()
        parts_v, parts_g = parts_v + v, parts_g + g
    assert abs(parts_v - v1) < 1e-12 * abs(v1) and np.abs(parts_g - g1).max() < 1e-10 * np.abs(g1).max()
    print("gradient kernel: %.2f ms for 1e6 tracks x %d directions" % (ms, len(names)))


def test_default_gradient_choice_follows_the_timing_probe(capsys):
    """``param_fitting(gradient=None)``: the optimiser gets the analytic gradient only where a gradient call is cheaper than the nvar + 1
    objective calls it replaces on THIS dataset (tracking._pick_gradient).  Two-state models (tangents in registers, xt_reg2.h) -> analytic; the
    3-state model at frame_len 6 (the reference's default window) -> analytic too since the reverse-mode kernels (xt_rev.h): same optimum as
    the finite-difference fit, in less time."""
    import time
    from extrack_amd import synth, tracking as T
    from test_hip_configs import _c3_tracks
    Cs = synth.brownian_tracks(200000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=3)
    p2 = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1], estimated_Fs=[0.5],
                           estimated_transition_rates=0.05)
    t0 = time.perf_counter()
    f2 = T.param_fitting({"30": Cs}, 0.02, params=p2, nb_states=2, frame_len=6, verbose=0, method="bfgs", cell_dims=[1])
    t_auto = time.perf_counter() - t0
    t0 = time.perf_counter()
    f2fd = T.param_fitting({"30": Cs}, 0.02, params=p2, nb_states=2, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
    t_fd = time.perf_counter() - t0
    tracks = _c3_tracks(3e4, seed0=900)
    p3 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                           estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.1)
    t0 = time.perf_counter()
    f3 = T.param_fitting(tracks, 0.02, params=p3, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1])
    t3_auto = time.perf_counter() - t0
    t0 = time.perf_counter()
    f3fd = T.param_fitting(tracks, 0.02, params=p3, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
    t3_fd = time.perf_counter() - t0
    capsys.readouterr()
    print("2 states: default fit %.2f s (ngev %d, nfev %d) vs fd fit %.2f s (nfev %d); 3 states F=6: default %.2f s (ngev %d, nfev %d) vs fd %.2f s (nfev %d)" % (
        t_auto, f2.ngev, f2.nfev, t_fd, f2fd.nfev, t3_auto, f3.ngev, f3.nfev, t3_fd, f3fd.nfev))
    assert f2.ngev > 0 and f2.residual[0] <= f2fd.residual[0] + 1e-6 * abs(f2fd.residual[0])
    assert t_auto < t_fd  # the probe's promise: never the slower way
    assert f3.ngev > 0 and f3.residual[0] <= f3fd.residual[0] + 1e-6 * abs(f3fd.residual[0])
    assert t3_auto < t3_fd


@pytest.mark.parametrize("S", [2, 3])
def test_gradient_more_track_lengths_than_one_launch_serves(S):
    """78 distinct track lengths (3 - 80): more than the 64 length buckets one launch serves, so the evaluation runs two launch groups whose
    {sum LL, gradient} are added on the device (refused with E_UNSUPPORTED until round 4).  Whole dataset = sum over the two halves of the
    buckets, value = extrack_loglik's; 2 states (tangents in registers) and 3 states (reverse mode)."""
    from extrack_amd import gradient, synth, tracking as T
    if S == 2:
        Ds, Tm, Fs = [0.0, 0.25], np.array([[0.9, 0.1], [0.1, 0.9]]), [0.6, 0.4]
        pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
    else:
        Ds, Tm, Fs = [0.0, 0.04, 0.25], np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]]), [0.3, 0.3, 0.4]
        pg = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3],
                               estimated_transition_rates=0.06)
    lst = [synth.brownian_tracks(40 + (L % 7) * 30, L, Ds, Tm, Fs, seed=300 + L) for L in range(3, 81)]
    names = gradient.free_names(pg)
    F = 5
    ts = T.TrackSet(lst)
    v, g = gradient.objective_and_gradient(pg, ts, 0.02, [1], S, 1, F, names=names)
    v0 = -ts.loglik(T._objective_model(pg, ts, 0.02, [1], None, S, 1, F, 1))
    ts.close()
    assert abs(v - v0) < 1e-12 * abs(v0), (v, v0)
    parts = []
    for half in (lst[:40], lst[40:]):
        ts = T.TrackSet(half, min_len=3, max_len=80)
        parts.append(gradient.objective_and_gradient(pg, ts, 0.02, [1], S, 1, F, names=names))
        ts.close()
    vs, gs = parts[0][0] + parts[1][0], parts[0][1] + parts[1][1]
    assert abs(vs - v) < 1e-11 * abs(v) and np.allclose(gs, g, rtol=1e-9, atol=1e-9 * np.abs(g).max()), (vs, v, np.abs(gs - g).max())
