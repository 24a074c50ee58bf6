"""GPU tests of the non-headline BASELINE.json configs at (or near) their stated sizes, through the C ABI:

  configs[2]  1e6 tracks, 3 states, mixed lengths 5-50, full param_fitting      -> test_c3_*
  configs[3]  1e7 tracks, 2 states, len 30 (8 GPUs)                             -> test_c4_* (one GPU: the whole 1e7, and its 8 shards)
  configs[4]  5e5 tracks, 4 states, len 60, nb_substeps 3 + predict_Bs          -> test_c5_*

Oracle comparisons run at sizes the CPU restatements finish in seconds (oracle_np for small slices, the plain-C restatement
oracle/extrack_oracle.c - pinned to the same golden vectors - for the larger ones); the full sizes are covered by
size-independent properties (total == sum of per-track values, additivity over shards, rows of posteriors sum to one).
Tolerances as everywhere: per-track LL abs 1e-10, totals rel 1e-12, posteriors abs 1e-9."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_LL = 1e-10
TOL_PRED = 1e-9

C3_DS = [0.0, 0.04, 0.25]
C3_T = np.array([[0.9, 0.1, 0.0], [0.05, 0.91, 0.04], [0.01, 0.06, 0.93]])  # Tutorial_ExTrack.ipynb:2440-2450
C3_F = [0.33, 0.33, 0.34]
C3_VALS = dict(D0=1e-4, D1=0.04, D2=0.25, LocErr=0.02, F0=0.33, F1=0.33, F2=0.34, p01=0.1, p02=0.001, p10=0.05, p12=0.04, p20=0.01, p21=0.06,
               pBL=0.01)
C5_DS = [0.0, 0.02, 0.1, 0.5]


def _params(vals):
    from extrack_amd.lmfit_compat import Parameters
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


def _c3_tracks(total, seed0=100):
    from extrack_amd import synth
    Tm = C3_T.copy()
    sizes = synth.bucket_sizes_geometric(int(total), list(range(5, 51)), 0.9)
    return {str(L): synth.brownian_tracks(n, L, C3_DS, Tm, C3_F, seed=seed0 + L) for L, n in sizes.items() if n > 0}


def _c5_vals():
    vals = dict(D0=1e-4, D1=0.02, D2=0.1, D3=0.5, LocErr=0.02, F0=.25, F1=.25, F2=.25, F3=.25, pBL=0.1)
    for i in range(4):
        for j in range(4):
            if i != j:
                vals["p%d%d" % (i, j)] = 0.05 / 3
    return vals


def _c5_tracks(n, seed=2):
    from extrack_amd import synth
    Tm = np.full((4, 4), 0.05 / 3)
    Tm[np.arange(4), np.arange(4)] = 0.95
    return synth.brownian_tracks(n, 60, C5_DS, Tm, [0.25] * 4, seed=seed)


def _oracle_c_bucket(Cs, vals, dt, cell_dims, ns, F, min_len, max_len, do_preds=False):
    from oracle import oracle_c, oracle_np as O
    LocErr, ds, Fs, T, pBL = O.extract_params(vals, dt, ns, 1)
    ps = O.p_stay_table(ds, len(ds), ns, cell_dims)
    return oracle_c.run(Cs, LocErr, ds, Fs, T, pBL, 0 if Cs.shape[1] == max_len else 1, ps, ns, F, min_len, do_preds=do_preds,
                        nthreads=min(16, os.cpu_count() or 1))


# ------------------------------------------------------------------------------------------------------------------
# configs[2]
# ------------------------------------------------------------------------------------------------------------------
def test_c3_mixed_lengths_objective_vs_oracle():
    """3 states, 46 buckets of lengths 5-50 (geometric sizes), 2e4 tracks: per-track LL of every bucket against the C restatement at
    frame_len 6 (the reference's default), and the objective against the numpy oracle at frame_len 4 on a 2e3-track dataset."""
    from extrack_amd import tracking as T
    from oracle import oracle_np as O
    tracks = _c3_tracks(2e4)
    assert len(tracks) == 46 and sorted(int(k) for k in tracks) == list(range(5, 51))
    p = _params(C3_VALS)
    _, lst, _ = T.engine.sort_buckets(tracks)
    ts = T.TrackSet(lst)
    model = T._objective_model(p, ts, 0.02, [1], None, 3, 1, 6, 1)
    tot, per = ts.loglik(model, per_track=True)
    ts.close()
    ref = np.concatenate([_oracle_c_bucket(b, C3_VALS, 0.02, [1], 1, 6, 5, 50)[0] for b in lst])
    assert np.abs(per - ref).max() < TOL_LL, np.abs(per - ref).max()
    assert abs(tot - ref.sum()) < 1e-12 * abs(tot)
    small = _c3_tracks(2e3, seed0=300)
    _, lst2, _ = T.engine.sort_buckets(small)
    got = T.cum_Proba_Cs(p, lst2, 0.02, [1], None, 3, 1, 4, verbose=0)
    want = O.cum_proba_cs(C3_VALS, small, 0.02, [1], None, 1, 4)
    assert abs(got - want) < 1e-12 * abs(want), (got, want)


def test_c3_gradient_reverse_vs_forward_mode(monkeypatch):
    """configs[2]-shaped data (46 buckets, 6e4 tracks, 3 states, 13 free parameters) at frame_len 6: the gradient of the fit objective from the
    reverse-mode kernels (the default for 3 states, xt_rev.h) against the forward-mode register kernels (xt_gradr.h) - two independent
    derivations of the same derivative - and the objective of both against the likelihood kernel."""
    from extrack_amd import gradient, tracking as T
    tracks = _c3_tracks(6e4, seed0=500)
    p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3],
                          estimated_transition_rates=0.06)
    names = gradient.free_names(p)
    assert len(names) == 13
    _, lst, _ = T.engine.sort_buckets(tracks)
    out = {}
    for path in ("auto", "gradr"):
        monkeypatch.setenv("EXTRACK_GRAD_PATH", path)
        ts = T.TrackSet(lst)
        try:
            out[path] = gradient.objective_and_gradient(p, ts, 0.02, [1], 3, 1, 6, names=names)
            out[path + "_lds"] = ts.ctx.last_launch_info()["lds_bytes"]
            ll = ts.loglik(T._objective_model(p, ts, 0.02, [1], None, 3, 1, 6, 1))
        finally:
            ts.close()
        assert abs(out[path][0] + ll) < 1e-12 * abs(ll)
    assert out["auto_lds"] != out["gradr_lds"]  # two different kernel families
    g0, g1 = out["auto"][1], out["gradr"][1]
    assert np.abs(g0 - g1).max() < 1e-10 * np.abs(g1).max(), (g0, g1)


def test_c3_full_size_properties():
    """configs[2] at full size (1e6 tracks, 3 states, 46 buckets, frame_len 6): total == sum of per-track values, additivity over
    two row shards with the dataset-global min/max length (what the multi-GPU path relies on), first 8 tracks of the shortest,
    a middle and the longest bucket against the numpy oracle."""
    from extrack_amd import tracking as T
    from oracle import oracle_np as O
    tracks = _c3_tracks(1e6)
    assert sum(len(v) for v in tracks.values()) == 1000000
    p = _params(C3_VALS)
    _, lst, _ = T.engine.sort_buckets(tracks)
    ts = T.TrackSet(lst)
    model = T._objective_model(p, ts, 0.02, [1], None, 3, 1, 6, 1)
    tot, per = ts.loglik(model, per_track=True)
    ts.close()
    assert np.all(np.isfinite(per)) and abs(tot - per.sum()) < 1e-12 * abs(tot)
    LocErr, ds, Fs, TT, pBL = O.extract_params(C3_VALS, 0.02, 1, 1)
    off = np.cumsum([0] + [len(b) for b in lst])
    for bi in (0, 20, 45):
        b = lst[bi]
        ref = O.proba_cs(b[:8], LocErr, ds, Fs, TT, pBL, 0 if b.shape[1] == 50 else 1, [1], 1, 6, 5)
        assert np.abs(per[off[bi]:off[bi] + 8] - ref).max() < TOL_LL
    parts = 0.0
    for half in (0, 1):
        sh = [b[:len(b) // 2] if half == 0 else b[len(b) // 2:] for b in lst]
        t2 = T.TrackSet([b for b in sh if len(b)], min_len=5, max_len=50)
        parts += t2.loglik(T._objective_model(p, t2, 0.02, [1], None, 3, 1, 6, 1))
        t2.close()
    assert abs(parts - tot) < 1e-12 * abs(tot)


def test_c3_three_state_fit_recovers_simulated_parameters(capsys):
    """Reduced-size configs[2] fit: 6e4 tracks of lengths 5-50 simulated from the tutorial's 3-state model
    (Tutorial_ExTrack.ipynb:2440-2450: D = 0 / 0.04 / 0.25, LocErr 0.02, fractions .33 / .33 / .34), param_fitting with 13 free
    parameters from generic starting values; the reference validates the same way (recovered vs simulated parameters)."""
    from extrack_amd import tracking as T
    tracks = _c3_tracks(6e4, seed0=500)
    p0 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                           estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.1)
    # frame_len 6: the reference's default (extrack/tracking.py:1304); gradient=None: the timing probe decides (the reverse-mode gradient on any dataset of this size; fit.gradient_path says which)
    fit = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, nb_substeps=1, frame_len=6, verbose=0, method="bfgs", cell_dims=[1])
    capsys.readouterr()
    v = {k: fit.params[k].value for k in fit.params}
    truth = T.cum_Proba_Cs(_params(dict(C3_VALS, pBL=0.0001)), T.engine.sort_buckets(tracks)[1], 0.02, [1], None, 3, 1, 6, verbose=0)
    capsys.readouterr()
    assert fit.residual[0] <= truth + 1e-6  # at least as good as (nearly) the generating parameters
    assert v["D0"] < 2e-3 and abs(v["D1"] - 0.04) < 0.006 and abs(v["D2"] - 0.25) < 0.02
    assert abs(v["LocErr"] - 0.02) < 0.002
    assert abs(v["F0"] - 0.33) < 0.06 and abs(v["F1"] - 0.33) < 0.06 and abs(v["F0"] + v["F1"] + v["F2"] - 1) < 1e-12
    rate = lambda pr: -np.log(1 - pr)  # Matrix_type 1: p = 1 - exp(-rate)
    assert abs(v["p01"] - rate(0.1)) < 0.03 and abs(v["p10"] - rate(0.05)) < 0.02 and abs(v["p21"] - rate(0.06)) < 0.02
    assert 10 < fit.nfev < 20000
    assert fit.gradient_path in ("analytic", "fd") and fit.gradient_why


def test_c3_full_size_fit_as_configs_2_states_it(capsys):
    """BASELINE configs[2] as written: the FULL param_fitting of 1e6 tracks (3 states, 46 buckets of lengths 5 - 50, 13 free parameters,
    frame_len 6) with default settings (gradient=None -> the probe picks the reverse-mode gradient): the optimum of the reference-style
    finite-difference fit to 1e-9 relative, within 120 objective + gradient calls - a regression of either turns this red."""
    from extrack_amd import tracking as T
    tracks = _c3_tracks(1e6, seed0=1000)
    p0 = T.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                           estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.1)
    fa = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1])
    ff = T.param_fitting(tracks, 0.02, params=p0, nb_states=3, frame_len=6, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
    capsys.readouterr()
    print("full C3 fit: default path %s: %d + %d calls -> %.6f ; fd: %d calls -> %.6f" % (fa.gradient_path, fa.nfev, fa.ngev, fa.residual[0], ff.nfev, ff.residual[0]))
    assert fa.gradient_path == "analytic", fa.gradient_why
    assert fa.residual[0] <= ff.residual[0] + 1e-9 * abs(ff.residual[0]), (fa.residual[0], ff.residual[0])
    assert fa.nfev + fa.ngev <= 120, (fa.nfev, fa.ngev)
    v = {k: fa.params[k].value for k in fa.params}
    assert abs(v["D1"] - 0.04) < 0.003 and abs(v["D2"] - 0.25) < 0.01 and abs(v["LocErr"] - 0.02) < 0.0005


# ------------------------------------------------------------------------------------------------------------------
# configs[3] on one GPU
# ------------------------------------------------------------------------------------------------------------------
def test_c4_ten_million_tracks_and_its_eight_shards():
    """configs[3]'s dataset (1e7 tracks x 30, 2 states) on ONE GPU: total == sum of per-track values, and the sum over the 8 row shards
    that 8 ranks would hold (1.25e6 each, dataset-global min/max length) equals the unsharded total - the only thing the 8-GPU
    run adds is the all-reduce of those 8 scalars."""
    from extrack_amd import synth, tracking as T
    from extrack_amd.distributed import shard_plan
    N, L = 10000000, 30
    Cs = synth.brownian_tracks(N, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=7)
    p = _params(dict(D0=0.0, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1))
    ts = T.TrackSet([Cs])
    model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
    tot, per = ts.loglik(model, per_track=True)
    ts.close()
    assert np.all(np.isfinite(per)) and abs(tot - per.sum()) < 1e-12 * abs(tot)
    plan = shard_plan([N], [L], 8)[0]
    assert [z - a for a, z in plan] == [1250000] * 8
    parts = 0.0
    for a, z in plan:
        t2 = T.TrackSet([Cs[a:z]], min_len=L, max_len=L)
        parts += t2.loglik(T._objective_model(p, t2, 0.02, [1], None, 2, 1, 6, 1))
        t2.close()
    assert abs(parts - tot) < 1e-12 * abs(tot)


# ------------------------------------------------------------------------------------------------------------------
# configs[4]
# ------------------------------------------------------------------------------------------------------------------
def test_c5_loglik_full_size_properties_and_oracle_slice():
    """configs[4] log-likelihood at full size: 5e5 tracks x 60, 4 states, nb_substeps 3, frame_len 4 (the final step of the reference
    enumerates 4^10 sequences per track): total == sum of per-track values, two-shard additivity, first 16 tracks against the C
    restatement (which materialises the 4^10 tail like the reference does)."""
    from extrack_amd import tracking as T
    Cs = _c5_tracks(500000)
    vals = _c5_vals()
    p = _params(vals)
    ts = T.TrackSet([Cs])
    model = T._objective_model(p, ts, 0.02, [1], None, 4, 3, 4, 1)
    tot, per = ts.loglik(model, per_track=True)
    ts.close()
    assert np.all(np.isfinite(per)) and abs(tot - per.sum()) < 1e-12 * abs(tot)
    ref = _oracle_c_bucket(Cs[:16], vals, 0.02, [1], 3, 4, 60, 60)[0]
    assert np.abs(per[:16] - ref).max() < TOL_LL, np.abs(per[:16] - ref).max()
    parts = 0.0
    for a, z in ((0, 250000), (250000, 500000)):
        t2 = T.TrackSet([Cs[a:z]], min_len=60, max_len=60)
        parts += t2.loglik(T._objective_model(p, t2, 0.02, [1], None, 4, 3, 4, 1))
        t2.close()
    assert abs(parts - tot) < 1e-12 * abs(tot)


def test_c5_predict_Bs_full_size():
    """configs[4] posterior annotation: predict_Bs (nb_substeps forced to 1, frame_len 5 = the reference's default) on 5e5 tracks x 60,
    4 states: shape, rows sum to one, first 16 and last 8 tracks against the numpy oracle, and the second half annotated alone
    (as another rank would) gives the same rows."""
    from extrack_amd import tracking as T
    from oracle import oracle_np as O
    Cs = _c5_tracks(500000)
    vals = _c5_vals()
    p = _params(vals)
    pr = T.predict_Bs({"60": Cs}, 0.02, p, cell_dims=[1], nb_states=4, frame_len=5)["60"]
    assert pr.shape == (500000, 60, 4)
    assert np.abs(pr.sum(-1) - 1).max() < 1e-12 and pr.min() >= 0
    ref = O.predict_bs(vals, {"60": np.concatenate([Cs[:16], Cs[-8:]])}, 0.02, [1], 5)["60"]
    assert np.abs(pr[:16] - ref[:16]).max() < TOL_PRED and np.abs(pr[-8:] - ref[16:]).max() < TOL_PRED
    half = T.predict_Bs({"60": Cs[250000:]}, 0.02, p, cell_dims=[1], nb_states=4, frame_len=5)["60"]
    assert np.abs(half - pr[250000:]).max() < 1e-12  # the posterior sums use LDS atomics: order-dependent in the last bits


def test_single_process_multi_device_entry_points():
    """extrack_multi_* (one process, several GPUs; include/extrack_hip.h): the row sharding, the concurrent per-device evaluations and the sum.
    This box has one GPU: the same device listed twice gives two shards with two contexts (RCCL refuses duplicate devices, so the totals are
    summed on the host - the path a node without librccl takes); with one device the communicator is not needed at all.  The 8-GPU RCCL path
    (ncclCommInitAll + one grouped ncclAllReduce per evaluation) is exercised by the driver's multi-GPU node only."""
    from extrack_amd import _lib, synth, tracking as T
    tracks = _c3_tracks(3000, seed0=40)
    _, lst, _ = T.engine.sort_buckets(tracks)
    ts = T.TrackSet(lst)
    model = T._objective_model(_params(C3_VALS), ts, 0.02, [1], None, 3, 1, 4, 1)
    ref = ts.loglik(model)
    for devs in ([0], [0, 0], [0, 0, 0]):
        mc = _lib.MultiContext(devs, use_rccl=1)
        try:
            for b in lst:
                mc.upload_bucket(b)
            got = mc.loglik(model)
            got2 = mc.loglik(model)
            assert not mc.uses_rccl
        finally:
            mc.close()
        assert got == got2 and abs(got - ref) < 1e-12 * abs(ref), (devs, got, ref)
    ts.close()
    with pytest.raises(_lib.ExtrackError):
        _lib.MultiContext([99])
