"""The N>1 path on CPU: world_size-2 gloo process group driving the PRODUCT call path -
``Comm.shard_trackset`` -> ``cum_Proba_Cs(comm=...)`` / ``param_fitting(comm=...)`` / ``predict_Bs(comm=...)`` - with ONE
substitution made in this test only: ``extrack_amd._lib.Context`` (the ctypes front of the HIP library; there is no GPU in
this container) is replaced by a stand-in that keeps the uploaded buckets on the host and evaluates them with the oracle.
Everything above the C ABI (sharding plan, dataset-global min/max length, empty-rank handling, chunk-aligned shards of the
threshold-fusion objective, the scalar all-reduce, the ordered gather of posterior rows, collective error agreement) is the
code that runs over RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELL = [1.0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleContext:
    """Test-only stand-in for extrack_amd._lib.Context: same methods, likelihoods from oracle/ (CPU)."""

    def __init__(self, device=0):
        self.device = int(device)
        self.buckets, self.data = [], []

    def upload_bucket(self, tracks, sigma=None):
        assert sigma is None
        self.buckets.append(tracks.shape + (0,))
        self.data.append(np.array(tracks, float))
        self.dts = getattr(self, "dts", {})
        return len(self.data) - 1

    def set_bucket_dt(self, bucket_id, dt):
        self.dts[bucket_id] = np.array(dt, float)

    def n_tracks(self):
        return sum(b[0] for b in self.buckets)

    def close(self):
        pass

    def set_stream(self, ptr):
        pass

    def _model(self, model):
        c = model.c
        le = np.array([c.locerr[i] for i in range(c.locerr_dims)])[None, None]
        return le, model.ds, model.Fs, model.TrMat, c.pBL, c.nb_substeps, c.frame_len, c.min_len, c.max_len

    def loglik(self, model, per_track=False):
        from oracle import oracle_np as O
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        per = [O.proba_cs(b, le, ds, Fs, T, pBL, 0 if b.shape[1] == hi else 1, CELL, ns, F, lo) for b in self.data]
        per = np.concatenate(per) if per else np.empty(0)
        return (per.sum(), per) if per_track else per.sum()

    def loglik_th(self, model, threshold=0.2, max_nb_states=120, chunk=2000, per_track=False):
        from oracle import oracle_th as OT
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        tot = 0.0
        for i, b in enumerate(self.data):
            for a0 in range(0, len(b), chunk):
                # per-track time steps: the model's ds are those of a unit time step, the reference's 3-D ds follow from the dt array
                dsc = ds if i not in self.dts else ds[None, None] * np.sqrt(self.dts[i][a0:a0 + chunk])[:, :, None]
                tot += OT.proba_cs_th(b[a0:a0 + chunk], le, dsc, Fs, T, pBL, 0 if b.shape[1] == hi else 1, CELL, ns, F, lo, threshold,
                                      max_nb_states).sum()
        return tot

    def loglik_grad(self, model, tangents):
        """sum LL and its derivative along every model tangent: central differences of the oracle (test stand-in; the kernel's own
        derivative is checked on the GPU, tests/test_hip_grad.py).  Additive over tracks, like the kernel's result."""
        from oracle import oracle_np as O
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)

        def total(ds_, Fs_, T_, pBL_, le_):
            return sum(O.proba_cs(b, le_, ds_, Fs_, T_, pBL_, 0 if b.shape[1] == hi else 1, CELL, ns, F, lo).sum() for b in self.data)
        h = 1e-6
        g = []
        if isinstance(tangents, dict):  # the packed form of gradient.model_tangents
            from extrack_amd import gradient
            tangents = gradient.tangent_rows(tangents)
        for t in tangents:
            def at(sg):
                return total(np.sqrt(ds ** 2 + sg * h * np.asarray(t["ds2"])), Fs + sg * h * np.asarray(t["Fs"]), T + sg * h * np.asarray(t["TrMat"]),
                             pBL + sg * h * t["pBL"], le + sg * h * np.asarray(t.get("locerr", 0.0)))
            g.append((at(+1) - at(-1)) / (2 * h))
        return total(ds, Fs, T, pBL, le), np.array(g)

    def loglik_th_grad(self, model, tangents, threshold=0.2, max_nb_states=120, chunk=2000):
        """Threshold-fusion sum LL and its derivative AT THE FROZEN PLAN of the evaluation along every model tangent: central differences of
        the oracle with every chunk's grouping held fixed (test stand-in for extrack_loglik_th_grad; the kernel's own derivative is checked
        in tests/test_emul_thgrad.py and on the GPU)."""
        from oracle import oracle_th as OT
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        chunks = [(b[a0:a0 + chunk], 0 if b.shape[1] == hi else 1) for b in self.data for a0 in range(0, len(b), chunk)]
        plans, base = [], 0.0
        for Cc, isBL in chunks:
            tr = []
            base += OT.proba_cs_th(Cc, le, ds, Fs, T, pBL, isBL, CELL, ns, F, lo, threshold, max_nb_states, trace=tr).sum()
            plans.append(tr)

        def total(ds_, Fs_, T_, pBL_, le_):
            return sum(OT.proba_cs_th(Cc, le_, ds_, Fs_, T_, pBL_, isBL, CELL, ns, F, lo, threshold, max_nb_states, plan=pl).sum()
                       for (Cc, isBL), pl in zip(chunks, plans))
        h = 1e-6
        if isinstance(tangents, dict):
            from extrack_amd import gradient
            tangents = gradient.tangent_rows(tangents)
        g = []
        for t in tangents:
            def at(sg):
                return total(np.sqrt(ds ** 2 + sg * h * np.asarray(t["ds2"])), Fs + sg * h * np.asarray(t["Fs"]), T + sg * h * np.asarray(t["TrMat"]),
                             pBL + sg * h * t["pBL"], le + sg * h * np.asarray(t.get("locerr", 0.0)))
            g.append((at(+1) - at(-1)) / (2 * h))
        return base, np.array(g)

    def segment_len_hist(self, model, bucket_id, max_nb_states=500):
        from oracle import oracle_hist as OH
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        b = self.data[bucket_id]
        return OH.p_segment_len(b, le, ds, Fs, T, lo, pBL, 0 if b.shape[1] == hi else 1, CELL, 1, max_nb_states)

    def predict_th(self, model, bucket_id, threshold=0.1, max_nb_states=200, nb_max=1):
        from oracle import oracle_th as OT
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        b = self.data[bucket_id]
        parts = []
        for a0 in range(0, len(b), nb_max):
            dsc = ds if bucket_id not in self.dts else ds[None, None] * np.sqrt(self.dts[bucket_id][a0:a0 + nb_max])[:, :, None]
            parts.append(OT.p_cs_inter_bound_stats_th(b[a0:a0 + nb_max], le, dsc, Fs, T, pBL, 0 if b.shape[1] == hi else 1, CELL, 1, F, 1, lo,
                                                      threshold, max_nb_states)[1])
        return np.concatenate(parts)

    def predict(self, model, bucket_id):
        from oracle import oracle_np as O
        le, ds, Fs, T, pBL, ns, F, lo, hi = self._model(model)
        b = self.data[bucket_id]
        return O.p_cs_inter_bound_stats(b, le, ds, Fs, T, pBL, 0 if b.shape[1] == hi else 1, CELL, 1, F, 1, lo)[1]


def _dataset():
    from extrack_amd import synth
    # the longest bucket has ONE track (one rank owns none of it); 6 small buckets of one 16-track chunk each
    lens = {5: 41, 9: 30, 12: 1, 6: 16, 7: 16, 8: 16, 10: 16, 11: 16}
    return {str(L): synth.brownian_tracks(n, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=L) for L, n in lens.items()}


def _worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    import contextlib
    import io
    import torch.distributed as dist
    from extrack_amd import _lib, tracking as T
    from extrack_amd.distributed import Comm, shard_plan, shard_range
    from extrack_amd.lmfit_compat import Parameters
    from oracle import oracle_np as O, oracle_th as OT
    _lib.Context = OracleContext  # the ONE substitution: no GPU here
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        comm = Comm()
        assert comm.backend == "gloo" and comm.world == world and comm.local_device() == 0
        vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
        p = Parameters()
        for k, v in vals.items():
            p.add(k, value=v)
        tracks = _dataset()
        keys, lst, _ = T.engine.sort_buckets(tracks)
        out = {"rank": rank}
        if scenario == "objective":
            # fixed-window objective through the product path
            ts = comm.shard_trackset(lst)
            assert (ts.min_len, ts.max_len) == (5, 12)
            out["has12"] = 12 in [s[1] for s in ts.shapes]  # one rank's local max length differs from the global one
            with contextlib.redirect_stdout(io.StringIO()):
                got = T.cum_Proba_Cs(p, ts, 0.02, CELL, None, 2, 1, 6, verbose=0, comm=comm)
                # a plain list is this rank's shard: the global min/max must still be agreed on
                loc, _ = comm.shard_buckets(lst)
                got_list = T.cum_Proba_Cs(p, loc, 0.02, CELL, None, 2, 1, 6, verbose=0, comm=comm)
            ts.close()
            out.update(got=got, got_list=got_list, ref=O.cum_proba_cs(vals, tracks, 0.02, CELL, None, 1, 6),
                       covered=comm.allreduce_scalar(ts.n_tracks, "sum"))
            # threshold-fusion objective: chunk-aligned shards -> the unsharded value
            chunk = 16
            ts = comm.shard_trackset(lst, chunk=chunk)
            with contextlib.redirect_stdout(io.StringIO()):
                th = T.cum_Proba_Cs(p, ts, 0.02, CELL, None, 2, 1, 6, verbose=0, threshold=0.2, max_nb_states=120,
                                    max_number_of_tracks_per_matrix=chunk, comm=comm, fusion="threshold")
            out.update(th=th, th_ref=OT.cum_proba_cs_th(vals, tracks, 0.02, CELL, None, 1, 6, 1, 0.2, 120, chunk=chunk), th_n=ts.n_tracks)
            ts.close()
            # the same with per-track time steps: the dt arrays are cut like the tracks, every chunk's field-of-view table comes from its own tracks
            rng = np.random.default_rng(3)
            dts = [0.02 * rng.uniform(0.5, 1.5, b.shape[:2]) for b in lst]
            ts = comm.shard_trackset(lst, chunk=chunk, dts=dts)
            assert ts.has_dt or ts.n_tracks == 0
            with contextlib.redirect_stdout(io.StringIO()):
                th_dt = T.cum_Proba_Cs(p, ts, None, CELL, None, 2, 1, 6, verbose=0, threshold=0.2, max_nb_states=120,
                                       max_number_of_tracks_per_matrix=chunk, comm=comm, fusion="threshold")
            out.update(th_dt=th_dt, th_dt_ref=OT.cum_proba_cs_th(vals, tracks, {k: d for k, d in zip(keys, dts)}, CELL, None, 1, 6, 1, 0.2, 120,
                                                                  chunk=chunk))
            ts.close()
            # objective + gradient: the (1 + nvar) vector is all-reduced; state-duration histograms: the small array is all-reduced
            from extrack_amd import gradient
            from extrack_amd.histograms import len_hist
            from oracle import oracle_hist as OH
            pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6],
                                   estimated_transition_rates=0.1)
            ts = comm.shard_trackset(lst)
            gv, gg = gradient.objective_and_gradient(pg, ts, 0.02, CELL, 2, 1, 4, comm=comm)
            ts.close()
            full = T.TrackSet(lst, device=0)  # the whole dataset on one stand-in context: what a single rank would compute
            gv1, gg1 = gradient.objective_and_gradient(pg, full, 0.02, CELL, 2, 1, 4)
            full.close()
            out.update(gv=gv, gg=gg.tolist(), gv1=gv1, gg1=gg1.tolist())
            # the same for the threshold-fusion objective (gradient at the frozen plan): chunk-aligned shards, one all-reduce of 1 + nvar doubles
            ts = comm.shard_trackset(lst, chunk=16)
            with contextlib.redirect_stdout(io.StringIO()):
                tv, tg = T.cum_Proba_Cs_grad(pg, gradient.free_names(pg), ts, 0.02, CELL, None, 2, 1, 4, verbose=0, threshold=0.2, max_nb_states=120,
                                             max_number_of_tracks_per_matrix=16, comm=comm, fusion="threshold")
                tv0 = T.cum_Proba_Cs(pg, ts, 0.02, CELL, None, 2, 1, 4, verbose=0, threshold=0.2, max_nb_states=120,
                                     max_number_of_tracks_per_matrix=16, comm=comm, fusion="threshold")
            ts.close()
            full = T.TrackSet(lst, device=0)
            tv1, tg1 = gradient.objective_and_gradient(pg, full, 0.02, CELL, 2, 1, 4, threshold_fusion=(0.2, 120, 16))
            full.close()
            out.update(tv=tv, tv0=tv0, tg=tg.tolist(), tv1=tv1, tg1=tg1.tolist())
            with contextlib.redirect_stdout(io.StringIO()):
                hh = len_hist(tracks, p, 0.02, cell_dims=CELL, nb_states=2, max_nb_states=30, comm=comm)
            out.update(hist_err=float(np.abs(hh - OH.len_hist(vals, tracks, 0.02, CELL, 30)).max()), hist_sum=float(hh.sum()))
            # posteriors: per-rank row ranges, ordered gather on rank 0
            pr = T.predict_Bs(tracks, 0.02, p, cell_dims=CELL, nb_states=2, frame_len=5, comm=comm)
            if rank == 0:
                ref = O.predict_bs(vals, tracks, 0.02, CELL, 5)
                out["pred_err"] = max(np.abs(pr[k] - ref[k]).max() for k in tracks)
                out["pred_keys"] = sorted(pr.keys(), key=int)
            else:
                assert pr is None
        elif scenario == "api":
            # the public API with a communicator where round 2 still refused: per-track time steps in param_fitting (fusion =
            # 'threshold') and threshold-fusion posteriors in chunks of nb_max > 1 tracks (shards = whole chunks)
            rng = np.random.default_rng(11)
            dtd = {k: 0.02 * rng.uniform(0.5, 1.5, v.shape[:2]) for k, v in tracks.items()}
            with contextlib.redirect_stdout(io.StringIO()):
                pr = T.predict_Bs(tracks, 0.02, p, cell_dims=CELL, nb_states=2, frame_len=4, threshold=0.2, max_nb_states=60, nb_max=7, comm=comm,
                                  fusion="threshold")
            if rank == 0:
                ref = OT.predict_bs_th(vals, tracks, 0.02, CELL, 4, 60, 0.2, None, 7)
                out["pred_th_err"] = max(np.abs(pr[k] - ref[k]).max() for k in tracks)
            else:
                assert pr is None
            pf = Parameters()
            for k, v in vals.items():
                pf.add(k, value=v, vary=(k in ("D1", "p01")), min=1e-4 if k != "F1" else -np.inf, max=1.0 if k != "F1" else np.inf)
            pf["F1"].expr = None
            with contextlib.redirect_stdout(io.StringIO()):
                fit = T.param_fitting(tracks, dtd, params=pf, nb_states=2, frame_len=4, verbose=0, method="powell", cell_dims=CELL, comm=comm,
                                      fusion="threshold", threshold=0.2, max_nb_states=60)
            out["fit_res"] = float(fit.residual[0])
            out["fit_ref"] = OT.cum_proba_cs_th({k: fit.params[k].value for k in fit.params}, tracks, dtd, CELL, None, 1, 4, 1, 0.2, 60, chunk=2000)
            out["fit_D1"] = fit.params["D1"].value
        elif scenario == "empty_rank":
            # fewer chunks than ranks: rank 1 holds nothing, contributes 0.0 and must not dead-lock the others
            small = {"9": tracks["9"][:10]}
            _, l2, _ = T.engine.sort_buckets(small)
            ts = comm.shard_trackset(l2, chunk=16)
            assert ts.n_tracks == (10 if rank == 0 else 0)
            with contextlib.redirect_stdout(io.StringIO()):
                th = T.cum_Proba_Cs(p, ts, 0.02, CELL, None, 2, 1, 6, verbose=0, max_number_of_tracks_per_matrix=16, comm=comm,
                                    fusion="threshold")
            ts.close()
            out.update(th=th, th_ref=OT.cum_proba_cs_th(vals, small, 0.02, CELL, None, 1, 6, 1, 0.2, 120, chunk=16))
            # the same with per-track time steps: the empty rank builds a model without a single chunk table (advisor r2: it used to
            # raise alone, before the collective, and leave rank 0 waiting)
            sdt = [0.02 * np.random.default_rng(5).uniform(0.5, 1.5, b.shape[:2]) for b in l2]
            ts = comm.shard_trackset(l2, chunk=16, dts=sdt)
            with contextlib.redirect_stdout(io.StringIO()):
                thd = T.cum_Proba_Cs(p, ts, None, CELL, None, 2, 1, 6, verbose=0, max_number_of_tracks_per_matrix=16, comm=comm,
                                     fusion="threshold")
            ts.close()
            out.update(thd=thd, thd_ref=OT.cum_proba_cs_th(vals, small, {"9": sdt[0]}, CELL, None, 1, 6, 1, 0.2, 120, chunk=16))
            # a failure inside ONE rank's evaluation is raised on every rank after the collective (nobody is left waiting)
            ts = comm.shard_trackset(l2, chunk=16)
            if rank == 0:
                def boom(*a, **k):
                    raise RuntimeError("injected failure on rank 0")
                ts.loglik_th = boom
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    T.cum_Proba_Cs(p, ts, 0.02, CELL, None, 2, 1, 6, verbose=0, max_number_of_tracks_per_matrix=16, comm=comm, fusion="threshold")
                out["agreed"] = False
            except RuntimeError as e:
                out["agreed"] = ("injected" in str(e)) if rank == 0 else ("other rank" in str(e))
            ts.close()
            # collective error agreement: an empty dataset raises on EVERY rank, nobody is left waiting in a collective
            try:
                comm.shard_trackset([np.zeros((0, 5, 2))])
                out["raised"] = False
            except ValueError:
                out["raised"] = True
        out["plan"] = shard_plan([41, 30, 1], [5, 9, 12], world)
        out["ranges"] = [shard_range(41, r, world) for r in range(world)]
        out["ranges_chunk"] = [shard_range(41, r, world, 16) for r in range(world)]
        q.put(out)
    finally:
        dist.destroy_process_group()


def _run(scenario):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r["rank"])


def test_two_rank_product_path_matches_unsharded():
    res = _run("objective")
    for r in res:
        assert r["covered"] == 152
        assert abs(r["got"] - r["ref"]) < 1e-12 * abs(r["ref"]), r
        assert abs(r["got_list"] - r["ref"]) < 1e-12 * abs(r["ref"]), r
        assert abs(r["th"] - r["th_ref"]) < 1e-12 * abs(r["th_ref"]), r
        assert abs(r["th_dt"] - r["th_dt_ref"]) < 1e-12 * abs(r["th_dt_ref"]), r
        assert abs(r["gv"] - r["gv1"]) < 1e-10 * abs(r["gv1"]) and np.allclose(r["gg"], r["gg1"], rtol=1e-6, atol=1e-4), (r["gg"], r["gg1"])
        assert abs(r["tv"] - r["tv1"]) < 1e-10 * abs(r["tv1"]) and abs(r["tv"] - r["tv0"]) < 1e-12 * abs(r["tv0"]), r
        assert np.allclose(r["tg"], r["tg1"], rtol=1e-6, atol=1e-4) and np.abs(r["tg"]).max() > 1.0, (r["tg"], r["tg1"])
        assert r["hist_err"] < 1e-10 and r["hist_sum"] > 1.0, r
        assert r["ranges"] == [(0, 21), (21, 41)] and r["ranges_chunk"] == [(0, 32), (32, 41)]
    assert res[0]["got"] == res[1]["got"] and res[0]["th"] == res[1]["th"]  # every rank sees the same reduced scalar
    assert sorted(r["has12"] for r in res) == [False, True]
    assert res[0]["pred_err"] < 1e-12 and res[0]["pred_keys"] == ["5", "6", "7", "8", "9", "10", "11", "12"]
    # chunk-aligned shards are balanced over ALL buckets (one-chunk buckets are dealt round-robin, not piled on rank 0)
    assert abs(res[0]["th_n"] - res[1]["th_n"]) <= 32 and min(res[0]["th_n"], res[1]["th_n"]) > 40


def test_public_api_with_communicator_dt_dict_and_chunked_threshold_posteriors():
    res = _run("api")
    r0 = [r for r in res if "pred_th_err" in r][0]
    assert r0["pred_th_err"] < 1e-12
    for r in res:
        assert abs(r["fit_res"] - r["fit_ref"]) < 1e-10 * abs(r["fit_ref"]), r
    assert res[0]["fit_res"] == res[1]["fit_res"] and res[0]["fit_D1"] == res[1]["fit_D1"]


def test_rank_without_tracks_and_collective_failure():
    res = _run("empty_rank")
    for r in res:
        assert abs(r["th"] - r["th_ref"]) < 1e-12 * abs(r["th_ref"]), r
        assert abs(r["thd"] - r["thd_ref"]) < 1e-12 * abs(r["thd_ref"]), r
        assert r["raised"] is True and r["agreed"] is True, r


def test_shard_range_partitions_exactly():
    from extrack_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_shard_plan_balances_many_small_buckets():
    """46 buckets of fewer than one 2000-track chunk each used to land on rank 0 entirely; equal buckets used to be
    ~2x unbalanced.  The plan must partition every bucket exactly, keep chunk alignment, and balance the work."""
    from extrack_amd.distributed import shard_plan
    lengths = list(range(5, 51))
    rng = np.random.default_rng(0)
    sizes = [int(x) for x in rng.integers(200, 1900, len(lengths))]
    plan = shard_plan(sizes, lengths, 8, chunk=2000)
    per_rank = [sum(z - a for a, z in (b[r] for b in plan)) for r in range(8)]
    assert sum(per_rank) == sum(sizes) and min(per_rank) > 0
    for n, b in zip(sizes, plan):
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(7))
        assert all(a % 2000 == 0 for a, z in b if z > a)
    work = [sum((z - a) * (L - 1) for (a, z), L in zip((b[r] for b in plan), lengths)) for r in range(8)]
    assert max(work) < 1.6 * (sum(work) / 8)
    # 1e6 tracks in 46 equal buckets, 2000-track chunks: balanced to a chunk overall
    sizes = [21740] * 46
    plan = shard_plan(sizes, lengths, 8, chunk=2000)
    per_rank = [sum(z - a for a, z in (b[r] for b in plan)) for r in range(8)]
    assert max(per_rank) - min(per_rank) <= 3 * 2000, per_rank
    # row-balanced mode: every bucket to +-1 row, totals to +-1 row as well (remainders rotate)
    plan = shard_plan([10, 10, 10, 10], [5, 5, 5, 5], 4)
    per_rank = [sum(z - a for a, z in (b[r] for b in plan)) for r in range(4)]
    assert per_rank == [10, 10, 10, 10]


def test_list_input_is_never_cached(monkeypatch):
    """the objective used to key a device-copy cache on id()/address of the caller's arrays, so an in-place
    edit (or a re-allocated array at the same address) silently evaluated stale device data.  List input is now uploaded per
    call: an edit must be seen, and nothing may stay alive behind the caller's back."""
    sys.path.insert(0, ROOT)
    from extrack_amd import _lib, synth, tracking as T
    from extrack_amd.lmfit_compat import Parameters
    from oracle import oracle_np as O
    created = []

    class Ctx(OracleContext):
        def __init__(self, device=0):
            super().__init__(device)
            created.append(self)
            self.closed = False

        def close(self):
            self.closed = True

    monkeypatch.setattr(_lib, "Context", Ctx)
    vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    a = synth.brownian_tracks(20, 7, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1)
    v1 = T.cum_Proba_Cs(p, [a], 0.02, CELL, None, 2, 1, 4, verbose=0)
    a[3, 2, 0] += 0.5  # in-place edit: same id, same address, same shape
    v2 = T.cum_Proba_Cs(p, [a], 0.02, CELL, None, 2, 1, 4, verbose=0)
    assert v1 != v2
    assert abs(v2 - O.cum_proba_cs(vals, {"7": a}, 0.02, CELL, None, 1, 4)) < 1e-12 * abs(v2)
    assert len(created) == 2 and all(c.closed for c in created)
    assert not hasattr(T, "_TRACKSET_CACHE")


def test_default_fusion_env_switch(monkeypatch):
    from extrack_amd import tracking as T
    monkeypatch.delenv("EXTRACK_FUSION", raising=False)
    assert T.default_fusion() == "window" and T._check_fusion(None) is False
    monkeypatch.setenv("EXTRACK_FUSION", "threshold")
    assert T._check_fusion(None) is True and T._check_fusion("window") is False
    monkeypatch.setenv("EXTRACK_FUSION", "bogus")
    with pytest.raises(ValueError):
        T._check_fusion(None)
