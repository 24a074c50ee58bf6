"""GPU tests of extrack_loglik_th_grad (csrc/xt_thgrad.h through the C ABI): the threshold-fusion objective that extrack.tracking.param_fitting
minimises in v1.6.3 (/root/reference/extrack/tracking.py:1371 -> :991 -> :427-743) AND its exact gradient at the frozen plan of the
evaluation.  Checker: Richardson-extrapolated central differences of the pinned oracle evaluated with the plan frozen (oracle_th ``plan=``).
Tolerances: value 1e-10 abs per track / 1e-12 rel on totals against extrack_loglik_th; gradient 1e-6 relative (measured ~1e-9)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _richardson(f, h):
    d1 = (f(h) - f(-h)) / (2 * h)
    d2 = (f(h / 2) - f(-h / 2)) / h
    return (4 * d2 - d1) / 3


def _dense_directions(rng, n, ds, Fs, Tm, LE, pBL, ns, cell_dims):
    """n random dense model directions (every field moves, components relative to the values) + the induced p_stay tangents, packed for
    Context.loglik_th_grad, and the per-direction perturbation dicts for the oracle."""
    from extrack_amd import engine
    S = len(ds)
    per_peak = LE.shape[1] != 1
    _, dps = engine.p_stay_table_grad(ds, S, ns, cell_dims)
    dirs = []
    for _ in range(n):
        d = dict(ds2=ds ** 2 * rng.uniform(-1, 1, S), Fs=Fs * rng.uniform(-1, 1, S), T=Tm * rng.uniform(-1, 1, (S, S)), pBL=pBL * rng.uniform(-1, 1))
        d["le"] = np.zeros(LE.shape[2]) if per_peak else LE[0, 0] * rng.uniform(-1, 1, LE.shape[2])
        dirs.append(d)
    t = dict(ds2=np.array([d["ds2"] for d in dirs]), Fs=np.array([d["Fs"] for d in dirs]), TrMat=np.array([d["T"] for d in dirs]),
             pBL=np.array([d["pBL"] for d in dirs]))
    t["p_stay"] = t["ds2"] @ dps.T
    if not per_peak:
        t["locerr"] = np.array([d["le"] for d in dirs])
    return dirs, t


def _check_case(T, OT, rng, Cs, LE, ds, Fs, Tm, pBL, isBL, cell_dims, ns, F, min_len, thr, mx, n_dir=3):
    """Returns (worst |LL - extrack_loglik_th|, worst relative gradient error) or None when the kernel does not serve the model."""
    from extrack_amd._lib import E_UNSUPPORTED, ExtrackError
    ts, le = T._one_bucket(Cs, LE, isBL, min_len, 0)
    try:
        model = ts.make_model(le, ds, Fs, Tm, pBL, cell_dims, ns, F)
        dirs, tang = _dense_directions(rng, n_dir, ds, Fs, Tm, LE, pBL, ns, cell_dims)
        chunk = max(len(Cs), 1)
        try:
            ll, g = ts.ctx.loglik_th_grad(model, tang, thr, mx, chunk)
        except ExtrackError as e:
            if e.code == E_UNSUPPORTED:
                return None
            raise
        ll0 = ts.loglik_th(model, thr, mx, chunk)
    finally:
        ts.close()
    tr = []
    OT.proba_cs_th(Cs, LE, ds, Fs, Tm, pBL, isBL, cell_dims, ns, F, min_len, thr, mx, trace=tr)
    f = lambda x, d: OT.proba_cs_th(Cs, LE if LE.shape[1] != 1 else LE + x * d["le"][None, None], np.sqrt(ds ** 2 + x * d["ds2"]), Fs + x * d["Fs"],
                                    Tm + x * d["T"], pBL + x * d["pBL"], isBL, cell_dims, ns, F, min_len, thr, mx, plan=tr).sum()
    fd = np.array([_richardson(lambda x: f(x, d), 2e-4) for d in dirs])
    return abs(ll - ll0) / max(abs(ll0), 1.0), float((np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())).max())


@pytest.mark.parametrize("kernel", ["1", "2"])
def test_th_grad_golden_ll_cases_vs_frozen_plan_differences(kernel, monkeypatch):
    """Both gradient kernels (EXTRACK_THG_KERNEL: 1 = one lane per track, csrc/xt_thgrad.h; 2 = one lane per (sequence, track), csrc/xt_thgrad2.h,
    where its tile fits - the launcher's own choice is 2 for up to 16 live sequences).  The 200 log-likelihood cases of the reference-generated fixture (2 - 4 states, nb_substeps 1 - 2, 2 - 25 positions, 1 - 60 tracks,
    scalar / per-peak errors, isBL 0 / 1, thresholds 0.05 - 0.5, max_nb_states 8 - 120): value = extrack_loglik_th's, gradient along three
    random dense model directions = the derivative of the oracle at the plan of the evaluation."""
    from extrack_amd import tracking as T
    from oracle import oracle_th as OT
    monkeypatch.setenv("EXTRACK_THG_KERNEL", kernel)
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases.npz"))
    rng = np.random.default_rng(2026)
    worst_v = worst_g = 0.0
    served = skipped = 0
    for row in meta:
        if row["do_preds"]:
            continue
        pre = "t%04d_" % row["id"]
        r = _check_case(T, OT, rng, data[pre + "Cs"], data[pre + "LE"], data[pre + "ds"], data[pre + "Fs"], data[pre + "T"], row["pBL"], row["isBL"],
                        row["cell_dims"], row["ns"], row["F"], row["min_len"], row["threshold"], row["max_nb_states"])
        if r is None:
            assert len(data[pre + "ds"]) ** (row["ns"] + 1) > 27, row  # only the models whose table adjoints exceed the LDS are refused
            skipped += 1
            continue
        assert r[0] < 1e-12 and r[1] < 1e-6, (row, r)
        worst_v, worst_g = max(worst_v, r[0]), max(worst_g, r[1])
        served += 1
    assert served + skipped == 200 and served >= 170, (served, skipped)
    print("th grad golden cases, kernel", kernel, ": served", served, "refused", skipped, "worst rel value diff", worst_v, "worst rel gradient error", worst_g)


@pytest.mark.parametrize("kernel", ["1", "2"])
def test_th_grad_extra_cases(kernel, monkeypatch):
    """Second fixture: 1-D / 3-D tracks, per-dimension and per-peak errors, nb_substeps up to 3, 5 states; both gradient kernels."""
    monkeypatch.setenv("EXTRACK_THG_KERNEL", kernel)
    from extrack_amd import tracking as T
    from oracle import oracle_th as OT
    meta = json.load(open(os.path.join(GOLDEN, "th_kernel_cases_extra.json")))
    data = np.load(os.path.join(GOLDEN, "th_kernel_cases_extra.npz"))
    rng = np.random.default_rng(7)
    served, worst = 0, 0.0
    for row in meta:
        if row.get("do_preds"):
            continue
        pre = "x%04d_" % row["id"]
        r = _check_case(T, OT, rng, data[pre + "Cs"], data[pre + "LE"], data[pre + "ds"], data[pre + "Fs"], data[pre + "T"], row["pBL"], row["isBL"],
                        row["cell_dims"], row["ns"], row["F"], row["min_len"], row["threshold"], row["max_nb_states"], n_dir=2)
        if r is None:
            continue
        assert r[0] < 1e-12 and r[1] < 1e-6, (row, r)
        served += 1
        worst = max(worst, r[1])
    assert served >= 30, served
    print("th grad extra cases served", served, "worst rel gradient error", worst)


def _c1():
    info = json.load(open(os.path.join(GOLDEN, "c1_simfov_10k.json")))
    data = np.load(os.path.join(GOLDEN, "c1_simfov_10k.npz"))
    return info, {k: data["tr_" + k] for k in info["keys"]}


def test_th_grad_parameter_level_on_c1_dataset_multi_bucket_multi_chunk():
    """cum_Proba_Cs_grad(fusion='threshold') on the sim_FOV dataset (16 buckets, several 500-track chunks per bucket, one launch): value =
    cum_Proba_Cs(fusion='threshold'), gradient with respect to the free PARAMETERS (host chain rule + kernel) against central differences of
    the objective itself (the plan may flip between the two evaluations of a difference: tolerance 1e-4)."""
    import contextlib
    import io
    from extrack_amd import gradient, tracking as T
    info, tracks = _c1()
    _, lst, _ = T.engine.sort_buckets(tracks)
    pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6],
                           estimated_transition_rates=0.1)
    names = gradient.free_names(pg)
    ts = T.TrackSet(lst)
    try:
        kw = dict(verbose=0, threshold=0.2, max_nb_states=120, max_number_of_tracks_per_matrix=500, fusion="threshold")
        with contextlib.redirect_stdout(io.StringIO()):
            v, g = T.cum_Proba_Cs_grad(pg, names, ts, 0.02, [1], None, 2, 1, 6, **kw)
            v2, g2 = T.cum_Proba_Cs_grad(pg, names, ts, 0.02, [1], None, 2, 1, 6, **kw)
            v0 = T.cum_Proba_Cs(pg, ts, 0.02, [1], None, 2, 1, 6, **kw)
            fd = []
            for nme in names:
                x0 = pg[nme].value
                h = 1e-5 * max(abs(x0), 1e-3)
                vals = []
                for sg in (+1, -1):
                    pg[nme].value = x0 + sg * h
                    pg.update_constraints()  # F1 = 1 - F0 follows
                    vals.append(T.cum_Proba_Cs(pg, ts, 0.02, [1], None, 2, 1, 6, **kw))
                pg[nme].value = x0
                pg.update_constraints()
                fd.append((vals[0] - vals[1]) / (2 * h))
    finally:
        ts.close()
    fd = np.array(fd)
    assert v == v2 and np.array_equal(g, g2)           # bit-reproducible
    assert abs(v - v0) < 1e-12 * abs(v0), (v, v0)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-4, list(zip(names, g, fd))


def test_th_fit_with_frozen_plan_gradient_beats_finite_differences():
    """param_fitting(fusion='threshold') - what v1.6.3 runs - with the analytic gradient: same optimum as the finite-difference fit (or
    better), far fewer objective calls, and the fit records which path it took."""
    import contextlib
    import io
    from extrack_amd import tracking as T
    info, tracks = _c1()
    fits = {}
    for grad in ("analytic", "fd", None):  # params=None: generate_params' defaults, D0 starts ON its lower bound
        with contextlib.redirect_stdout(io.StringIO()):
            fits[grad] = T.param_fitting(tracks, 0.02, nb_states=2, frame_len=6, cell_dims=[1], verbose=0, fusion="threshold", gradient=grad)
    fa, ff, fn = fits["analytic"], fits["fd"], fits[None]
    assert fa.gradient_path == "analytic" and ff.gradient_path == "fd" and fn.gradient_path in ("analytic", "fd") and fn.gradient_why
    assert fa.residual[0] <= ff.residual[0] + 1e-7 * abs(ff.residual[0]), (fa.residual[0], ff.residual[0])
    assert fa.nfev + getattr(fa, "ngev", 0) < 0.5 * ff.nfev, (fa.nfev, getattr(fa, "ngev", 0), ff.nfev)
    for k in ("D1", "LocErr", "F0"):
        assert abs(fa.params[k].value - ff.params[k].value) < 2e-2 * abs(ff.params[k].value), (k, fa.params[k].value, ff.params[k].value)
    print("threshold-fusion fit: analytic %d + %d calls -> %.6f ; fd %d calls -> %.6f ; default path %s (%s)"
          % (fa.nfev, getattr(fa, "ngev", 0), fa.residual[0], ff.nfev, ff.residual[0], fn.gradient_path, fn.gradient_why))


def test_th_grad_three_states_mixed_lengths_shard_additivity():
    """C3-shaped data (3 states, lengths 5 - 40, 2000-track chunks): the gradient of the whole dataset = the sum over two chunk-aligned
    shards (what two ranks would all-reduce), and the objective value is extrack_loglik_th's."""
    from extrack_amd import gradient, synth, tracking as T
    Ds, Tm, Fs = [0.0, 0.04, 0.25], np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]]), [0.3, 0.3, 0.4]
    sizes = synth.bucket_sizes_geometric(60000, list(range(5, 41)), 0.9)
    lst = [synth.brownian_tracks(n, L, Ds, Tm, Fs, seed=100 + L) for L, n in sizes.items() if n > 0]
    pg = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3],
                           estimated_transition_rates=0.06)
    names = gradient.free_names(pg)
    tf = (0.2, 120, 2000)
    lo, hi = min(b.shape[1] for b in lst), max(b.shape[1] for b in lst)
    ts = T.TrackSet(lst)
    v, g = gradient.objective_and_gradient(pg, ts, 0.02, [1], 3, 1, 6, names=names, threshold_fusion=tf)
    model = T._objective_model(pg, ts, 0.02, [1], None, 3, 1, 6, 1)
    v0 = -ts.loglik_th(model, *tf)
    ts.close()
    assert abs(v - v0) < 1e-12 * abs(v0), (v, v0)
    parts = []
    for half in (0, 1):
        sh = [b[:(len(b) // 4000) * 2000] if half == 0 else b[(len(b) // 4000) * 2000:] for b in lst]
        sh = [b for b in sh if len(b)]
        ts = T.TrackSet(sh, min_len=lo, max_len=hi)
        parts.append(gradient.objective_and_gradient(pg, ts, 0.02, [1], 3, 1, 6, names=names, threshold_fusion=tf))
        ts.close()
    vs, gs = parts[0][0] + parts[1][0], parts[0][1] + parts[1][1]
    assert abs(vs - v) < 1e-11 * abs(v) and np.allclose(gs, g, rtol=1e-9, atol=1e-9 * np.abs(g).max()), (vs, v)
