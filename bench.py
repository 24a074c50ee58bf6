#!/usr/bin/env python3
"""Headline benchmark: log-likelihood evaluations per second on synthetic track batches.

A "step" is one evaluation of -sum(LL) over the whole dataset (local kernel on every rank + one all-reduce of the fp64 scalar over
RCCL); `value` is in units of "1e6-track log-likelihood evaluations per second" (whole job, all GPUs).

Workloads (all BASELINE.json shapes: 2 states, length 30, 2-D, nb_substeps=1, frame_len=6, resident in HBM before the timed region):
  c2   configs[1], 1e6 tracks PER GPU (weak scaling)              - the headline on ONE GPU
  c2s  configs[1], 1e6 tracks IN TOTAL, row-sharded (strong)      - the headline on N > 1 GPUs (what BASELINE.json's metric names)
  c4   configs[3], 1e7 tracks in total, row-sharded (strong)
Without ``--config``, an N > 1 run measures all three back to back (`scaling_runs`), each with ms_per_step, kernel_ms and
comm_ms = step - kernel; with ``--config`` only that one.  At N = 1 an `extra` block (outside the timed region) reports the other
BASELINE configs on the same GPU: configs[2] (3 states, 46 length buckets), configs[4] (4 states, nb_substeps 3) and its predict_Bs
annotation, configs[3] on one GPU, each with algorithmic bytes and the flop model of SURVEY.md section 8(d).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            # starts its own N rank processes (one per GPU) and relays rank 0's JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETTLE_LAUNCHES = 24  # untimed launches before the warm-up steps of a workload (GPU clock ramp of a fresh process)
N_TRACKS, LEN, DIMS, S, NS, FRAME = int(os.environ.get("EXTRACK_BENCH_N_TRACKS", "1000000")), 30, 2, 2, 1, 6  # (the env override: CPU rehearsal sizes)
DS_COEF, TRMAT, FS, LOCERR, DT, PBL, CELL = [0.0, 0.25], [[0.9, 0.1], [0.1, 0.9]], [0.6, 0.4], 0.02, 0.02, 0.1, [1.0]
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def cpu_baseline_worker(args):
    from oracle import oracle_np as O
    Cs, model = args
    t0 = time.perf_counter()
    O.proba_cs(Cs, *model)
    return time.perf_counter() - t0


def _one_socket_cores():
    """Physical cores of one socket (the north-star baseline is "single-socket numpy"), capped by the affinity mask."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        phys = set()
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        sockets = len(set(p for p, _ in phys)) or 1
        per_socket = max(1, len(phys) // sockets)
        return max(1, min(per_socket, avail))
    except Exception:
        return avail


def cpu_baseline(sample_tracks_per_core=8000):
    """numpy port (oracle/oracle_np.py, parity-pinned to the reference) on a bounded sample, one socket's cores."""
    import multiprocessing as mp
    from extrack_amd import synth
    cores = _one_socket_cores()
    n = sample_tracks_per_core * cores
    Cs = synth.brownian_tracks(n, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=123)
    ds = np.sqrt(2 * np.array(DS_COEF) * DT)
    T = 1 - np.exp(-np.array(TRMAT)); T[np.arange(S), np.arange(S)] = 0; T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    model = (np.array([[[LOCERR]]]), ds, np.array(FS), T, PBL, 0, CELL, NS, FRAME, LEN)
    chunks = [(Cs[a:a + 2000], model) for a in range(0, n, 2000)]   # 2000 = the reference's chunk (tracking.py:991)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        pool.map(cpu_baseline_worker, chunks)
    wall = time.perf_counter() - t0
    tps = n / wall
    return {"value": tps / N_TRACKS, "unit": "1e6-track LL evals/s", "cores": cores, "kind": "port",
            "sample": "%d tracks (len %d, 2 states, frame_len %d) in 2000-track chunks over a %d-process fork pool, "
                      "numpy oracle; %.0f tracks/s, scaled linearly to 1e6 tracks" % (n, LEN, FRAME, cores, tps),
            "tracks_per_s": tps}


def cpu_th_worker(args):
    from oracle import oracle_th as OT
    Cs, model = args
    t0 = time.perf_counter()
    OT.proba_cs_th(Cs, *model)
    return time.perf_counter() - t0


def cpu_baseline_th(cores, sample_tracks_per_core=16000):
    """numpy port of the threshold-fusion kernel (oracle/oracle_th.py, pinned to reference fixtures), 2000-track chunks."""
    import multiprocessing as mp
    from extrack_amd import synth
    n = sample_tracks_per_core * cores
    Cs = synth.brownian_tracks(n, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=77)
    ds = np.sqrt(2 * np.array(DS_COEF) * DT)
    T = 1 - np.exp(-np.array(TRMAT)); T[np.arange(S), np.arange(S)] = 0; T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    model = (np.array([[[LOCERR]]]), ds, np.array(FS), T, PBL, 0, CELL, NS, FRAME, LEN, 0.2, 120)
    chunks = [(Cs[a:a + 2000], model) for a in range(0, n, 2000)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(cpu_th_worker, chunks)
    wall = time.perf_counter() - t0
    return {"value": n / wall / N_TRACKS, "unit": "1e6-track LL evals/s", "cores": cores, "kind": "port",
            "sample": "%d tracks in 2000-track chunks, numpy oracle_th; %.0f tracks/s" % (n, n / wall)}


def cpu_hist_worker(args):
    from oracle import oracle_hist as OH
    tr, vals = args
    return OH.len_hist(vals, tr, DT, cell_dims=CELL, max_nb_states=500).sum()


def cpu_refine_worker(args):
    from oracle import oracle_refine as OR
    tr, dsr = args
    return OR.position_refinement(tr, LOCERR, dsr, np.array(FS), np.array(TRMAT), 6, 0.1, 1000)[0]["30"].sum()


def cpu_baseline_compiled(cores, n=400000):
    """Secondary CPU number: the plain-C restatement (oracle/extrack_oracle.c, gcc -O2 -fopenmp, log domain like the reference)
    on the same socket.  Reported next to the numpy baseline so that the GPU/CPU ratio can also be read against compiled code."""
    from extrack_amd import synth
    from oracle import oracle_c, oracle_np as O
    Cs = synth.brownian_tracks(n, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=321)
    ds = np.sqrt(2 * np.array(DS_COEF) * DT)
    T = 1 - np.exp(-np.array(TRMAT)); T[np.arange(S), np.arange(S)] = 0; T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    ps = O.p_stay_table(ds, S, NS, CELL)
    oracle_c.run(Cs[:2000], np.array([[[LOCERR]]]), ds, FS, T, PBL, 0, ps, NS, FRAME, LEN, nthreads=cores)
    t0 = time.perf_counter()
    oracle_c.run(Cs, np.array([[[LOCERR]]]), ds, FS, T, PBL, 0, ps, NS, FRAME, LEN, nthreads=cores)
    wall = time.perf_counter() - t0
    return {"value": n / wall / N_TRACKS, "unit": "1e6-track LL evals/s", "cores": cores, "kind": "port (C, OpenMP)",
            "sample": "%d tracks, %.0f tracks/s" % (n, n / wall)}


def _flop_model(lengths_counts, S, F, ns, D):
    """Algorithmic fp64 flops of one evaluation, SURVEY.md section 8(d): per track (L-2) * S^F * [(4D+8) + 3S + S(D+4) + 4 flops
    + (S exp + 2 log + 2 div) at ~20 flops each] (the reference's operation count per kept sequence and step)."""
    per_entry = (4 * D + 8) + 3 * S + S * (D + 4) + 4 + (S + 4) * 20
    return float(sum(n * max(L - 2, 0) for L, n in lengths_counts)) * (S ** F) * per_entry


def other_configs(device, no_cpu_baseline=False, no_fits=False):
    """configs[2] and configs[4] on the same GPU (N = 1 only, outside the timed region): ms per evaluation / tracks per second with
    the HIP-event kernel time, algorithmic bytes (one read of the tracks; posteriors add their output) and the flop model."""
    import torch
    from extrack_amd import synth, tracking
    from extrack_amd.lmfit_compat import Parameters

    def P(vals):
        p = Parameters()
        for k, v in vals.items():
            p.add(k, value=v)
        return p

    def timed(fn, n, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ks = []
        for _ in range(n):
            r = fn()
            ks.append(ts.ctx.last_kernel_ms())
        return (time.perf_counter() - t0) / n, float(np.mean(ks)), r

    out = {}
    # ---- configs[2]: 1e6 tracks, 3 states, lengths 5..50 (46 buckets, geometric sizes), one objective evaluation
    Ds = [0.0, 0.04, 0.25]
    Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    sizes = synth.bucket_sizes_geometric(N_TRACKS, list(range(5, 51)), 0.9)
    tracks = {str(L): synth.brownian_tracks(n, L, Ds, Tm, [0.3, 0.3, 0.4], seed=1000 + L) for L, n in sizes.items() if n > 0}
    vals = dict(D0=1e-4, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.07, p02=0.03, p10=0.05, p12=0.05, p20=0.03, p21=0.07,
                pBL=0.1)
    _, lst, _ = tracking.engine.sort_buckets(tracks)
    ts = tracking.TrackSet(lst, device=device)
    lc = [(b.shape[1], len(b)) for b in lst]
    nbytes = sum(b.nbytes for b in lst)
    del tracks, lst
    for F in (6, 4):
        model = tracking._objective_model(P(vals), ts, DT, CELL, None, 3, 1, F, 1)
        wall, kms, v = timed(lambda: ts.loglik(model), 5)
        fl = _flop_model(lc, 3, F, 1, DIMS)
        out["c3_loglik_F%d" % F] = {"what": "configs[2]: 1e6 tracks, 3 states, 46 buckets len 5-50, frame_len %d, one evaluation" % F,
                                    "ms_per_eval": wall * 1e3, "kernel_ms": kms, "algorithmic_bytes": nbytes,
                                    "hbm_gbs": nbytes / (kms * 1e-3) / 1e9, "hbm_frac": nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "flop_model": fl, "fp64_tflops": fl / (kms * 1e-3) / 1e12,
                                    "fp64_frac": fl / (kms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, "neg_loglik": -v,
                                    "launch": ts.ctx.last_launch_info()}
    # likelihood + exact gradient in one pass (13 free parameters) vs the 14 evaluations a finite-difference gradient costs
    from extrack_amd import gradient
    pg = tracking.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3],
                                  estimated_transition_rates=0.06)
    names = gradient.free_names(pg)
    for _ in range(2):
        gv, gg = gradient.objective_and_gradient(pg, ts, DT, CELL, 3, 1, 4, names=names)
    out["c3_loglik_grad_F4"] = {"what": "configs[2], frame_len 4: -sum(LL) AND its exact gradient (13 free parameters) by the reverse-mode kernel "
                                        "(xt_rev.h: one forward + one backward sweep)", "kernel_ms": ts.ctx.last_grad_ms(),
                                "n_directions": len(names), "fd_equivalent_ms": (len(names) + 1) * out["c3_loglik_F4"]["kernel_ms"],
                                "grad_inf_norm": float(np.abs(gg).max())}
    for _ in range(2):
        gv, gg = gradient.objective_and_gradient(pg, ts, DT, CELL, 3, 1, 6, names=names)
    step_b = 243 * 36  # merged-state log of a step: 243 groups x (W mantissa, exponent, mean[2], variance)
    log_bytes = int(sum((L - 2) * n for L, n in lc) * step_b)
    out["c3_loglik_grad_F6"] = {"what": "configs[2], frame_len 6 (the reference's default): the same gradient by the reverse-mode kernel - its cost does not depend on "
                                        "the number of parameters; the forward-mode kernels (xt_gradr.h, r03: 600 ms) are kept for EXTRACK_GRAD_PATH=gradr",
                                "kernel_ms": ts.ctx.last_grad_ms(), "n_directions": len(names),
                                "fd_equivalent_ms": (len(names) + 1) * out["c3_loglik_F6"]["kernel_ms"],
                                "algorithmic_bytes": nbytes + 2 * log_bytes,
                                "byte_model": "tracks once + the merged-state log written by the forward sweep and read back by the backward sweep",
                                "hbm_gbs": (nbytes + 2 * log_bytes) / (ts.ctx.last_grad_ms() * 1e-3) / 1e9}
    model = tracking._objective_model(P(vals), ts, DT, CELL, None, 3, 1, 6, 1)
    # two untimed evaluations: the first learns the sequence counts (LDS sizing), the second allocates the second stream's launch buffers
    wall, kms, v = timed(lambda: ts.loglik_th(model, 0.2, 120, 2000), 5, warm=2)
    ts.th_freeze_plan(True)
    wallz, kmsz, vz = timed(lambda: ts.loglik_th(model, 0.2, 120, 2000), 5, warm=1)
    ts.th_freeze_plan(False)
    out["c3_loglik_threshold"] = {"what": "configs[2] through the threshold-fusion kernels (v1.6.3 defaults, frame_len 6)", "ms_per_eval": wall * 1e3,
                                  "kernel_ms": kms, "algorithmic_bytes": nbytes, "hbm_gbs": nbytes / (kms * 1e-3) / 1e9, "neg_loglik": -v,
                                  "frozen_plan_ms_per_eval": wallz * 1e3, "frozen_plan_kernel_ms": kmsz, "frozen_plan_same_value": bool(vz == v)}
    # ... and its exact gradient at the frozen plan (extrack_loglik_th_grad, round 4): what an optimiser iteration of the v1.6.3 objective costs
    for _ in range(2):
        gv, gg = gradient.objective_and_gradient(pg, ts, DT, CELL, 3, 1, 6, names=names, threshold_fusion=(0.2, 120, 2000))
    out["c3_loglik_th_grad"] = {"what": "configs[2] through the threshold-fusion kernels: -sum(LL) AND its exact gradient (13 free parameters) at the frozen plan of the "
                                        "evaluation (plan kernel + xt_thg_kernel: one forward + one backward sweep, one lane per track - ~33 live sequences per step)",
                                "kernel_ms": ts.ctx.last_grad_ms(), "n_directions": len(names),
                                "fd_equivalent_ms": (len(names) + 1) * out["c3_loglik_threshold"]["kernel_ms"], "grad_inf_norm": float(np.abs(gg).max()),
                                "launch": ts.ctx.last_launch_info()}
    ts.close()
    # configs[2] as BASELINE states it: the FULL fit of the 1e6-track dataset (3 states, 13 free parameters, frame_len 6), default settings -
    # param_fitting decides by a timing probe whether the optimiser gets the one-pass gradient (ngev > 0) or differences the objective
    import contextlib
    import io
    tracks = {str(L): synth.brownian_tracks(n, L, Ds, Tm, [0.3, 0.3, 0.4], seed=1000 + L) for L, n in sizes.items() if n > 0}
    p0 = tracking.generate_params(nb_states=3, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.02, 0.4],
                                  estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.1)
    fits3 = {}
    for grad in (() if no_fits else (None, "fd")):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            r = tracking.param_fitting(tracks, DT, params=p0, nb_states=3, frame_len=6, cell_dims=CELL, verbose=0, device=device, gradient=grad)
            t_fit = time.perf_counter() - t0
        fits3["default" if grad is None else grad] = {"seconds": t_fit, "objective_calls": int(r.nfev), "gradient_calls": int(getattr(r, "ngev", 0)),
                                                       "neg_loglik": float(r.residual[0]),
                                                       "fitted": {k: float(r.params[k].value) for k in ("D1", "D2", "LocErr", "F0", "F1")}}
    if fits3:
        out["c3_full_fit_F6"] = {"what": "configs[2]: param_fitting on 1e6 tracks, 3 states, 46 buckets, frame_len 6, BFGS from a generic start (incl. the upload); "
                                         "'default' = gradient=None (timing probe -> reverse-mode gradient), 'fd' = finite differences like the reference",
                                 "seconds": fits3["default"]["seconds"], "objective_calls": fits3["default"]["objective_calls"],
                                 "gradient_calls": fits3["default"]["gradient_calls"], "neg_loglik": fits3["default"]["neg_loglik"], "fit": fits3,
                                 "simulated": {"D1": 0.04, "D2": 0.25, "LocErr": LOCERR, "F0": 0.3, "F1": 0.3}}
    # the same fit of the objective extrack.tracking.param_fitting minimises in v1.6.3 (threshold fusion): default = the frozen-plan driver
    # (plan, minimise at that plan with the exact gradient, re-plan), fd = the reference's finite-difference BFGS
    fits3t = {}
    for grad in (() if no_fits else (None, "fd")):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            r = tracking.param_fitting(tracks, DT, params=p0, nb_states=3, frame_len=6, cell_dims=CELL, verbose=0, device=device, gradient=grad, fusion="threshold")
            t_fit = time.perf_counter() - t0
        fits3t["default" if grad is None else grad] = {"seconds": t_fit, "objective_calls": int(r.nfev), "gradient_calls": int(getattr(r, "ngev", 0)),
                                                        "neg_loglik": float(r.residual[0]), "gradient_path": getattr(r, "gradient_path", None),
                                                        "gradient_why": getattr(r, "gradient_why", None), "plan_rounds": int(getattr(r, "plan_rounds", 0)),
                                                        "fitted": {k: float(r.params[k].value) for k in ("D1", "D2", "LocErr", "F0", "F1")}}
    if fits3t:
        out["c3_full_fit_threshold"] = {"what": "configs[2] with fusion='threshold' (v1.6.3's objective): param_fitting on 1e6 tracks, 3 states, 46 buckets, frame_len 6 from "
                                                "the same generic start; 'default' = gradient=None (probe -> frozen-plan gradient, plan / minimise / re-plan rounds), 'fd' = "
                                                "finite differences of the re-planning objective like the reference",
                                        "seconds": fits3t["default"]["seconds"], "objective_calls": fits3t["default"]["objective_calls"],
                                        "gradient_calls": fits3t["default"]["gradient_calls"], "neg_loglik": fits3t["default"]["neg_loglik"], "fit": fits3t}
    del tracks
    # the same for the headline dataset (configs[1]: 1e6 x 30, 2 states, 7 free parameters): analytic gradient vs finite differences
    c2 = {str(LEN): synth.brownian_tracks(N_TRACKS, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=0)}
    p2 = tracking.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1], estimated_Fs=[0.5],
                                  estimated_transition_rates=0.05)
    fits2 = {}
    for grad in (() if no_fits else (None, "fd")):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            r = tracking.param_fitting(c2, DT, params=p2, nb_states=2, frame_len=6, cell_dims=CELL, verbose=0, device=device, gradient=grad)
            fits2["default" if grad is None else grad] = {"seconds": time.perf_counter() - t0, "objective_calls": int(r.nfev),
                                                           "gradient_calls": int(getattr(r, "ngev", 0)), "neg_loglik": float(r.residual[0])}
    # the threshold-fusion objective of the same dataset and its exact gradient at the frozen plan (12 live sequences: csrc/xt_thgrad2.h)
    try:
        ts = tracking.TrackSet([c2[str(LEN)]], device=device)
        pg2 = tracking.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[LOCERR], estimated_Fs=[0.6],
                                       estimated_transition_rates=0.1)
        names2 = gradient.free_names(pg2)
        m2 = tracking._objective_model(pg2, ts, DT, CELL, None, 2, 1, 6, 1)
        w2, k2, v2 = timed(lambda: ts.loglik_th(m2, 0.2, 120, 2000), 5, warm=2)
        for _ in range(3):
            gv2, gg2 = gradient.objective_and_gradient(pg2, ts, DT, CELL, 2, 1, 6, names=names2, threshold_fusion=(0.2, 120, 2000))
        out["c2_loglik_th_grad"] = {"what": "configs[1] data through the threshold-fusion kernels: -sum(LL) AND its exact gradient (7 free parameters) at the frozen plan "
                                            "(plan kernel + xt_thg2_kernel: lanes over (sequence, track), live state in LDS)",
                                    "kernel_ms": ts.ctx.last_grad_ms(), "n_directions": len(names2), "objective_kernel_ms": k2,
                                    "fd_equivalent_ms": (len(names2) + 1) * k2, "same_value": bool(abs(gv2 + v2) <= 1e-12 * abs(v2)), "launch": ts.ctx.last_launch_info()}
        ts.close()
    except Exception as e:  # noqa: BLE001
        out["c2_loglik_th_grad"] = {"error": str(e)}
    if fits2:
        out["c2_full_fit_F6"] = {"what": "configs[1] data: param_fitting on 1e6 x 30, 2 states, frame_len 6 from a generic start; 'default' = gradient=None "
                                     "(timing probe -> one-pass analytic gradient, tangents in registers), 'fd' = finite differences like the reference", "fit": fits2}
    del c2
    # ---- configs[4]: 5e5 tracks x 60, 4 states, nb_substeps 3 (frame_len 4) + predict_Bs (nb_substeps 1, frame_len 5)
    N5, L5 = 500000, 60
    Tm = np.full((4, 4), 0.05 / 3)
    Tm[np.arange(4), np.arange(4)] = 0.95
    Cs = synth.brownian_tracks(N5, L5, [0.0, 0.02, 0.1, 0.5], Tm, [0.25] * 4, seed=2)
    vals = dict(D0=1e-4, D1=0.02, D2=0.1, D3=0.5, LocErr=0.02, F0=.25, F1=.25, F2=.25, F3=.25, pBL=0.1)
    for i in range(4):
        for j in range(4):
            if i != j:
                vals["p%d%d" % (i, j)] = 0.05 / 3
    ts = tracking.TrackSet([Cs], device=device)
    nbytes = Cs.nbytes
    del Cs
    model = tracking._objective_model(P(vals), ts, DT, CELL, None, 4, 3, 4, 1)
    wall, kms, v = timed(lambda: ts.loglik(model), 3)
    fl = _flop_model([(L5, N5)], 4, 4, 3, DIMS)
    out["c5_loglik_ns3_F4"] = {"what": "configs[4]: 5e5 tracks x 60, 4 states, nb_substeps 3, frame_len 4, one evaluation", "ms_per_eval": wall * 1e3,
                               "kernel_ms": kms, "algorithmic_bytes": nbytes, "hbm_gbs": nbytes / (kms * 1e-3) / 1e9,
                               "hbm_frac": nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "flop_model": fl, "fp64_tflops": fl / (kms * 1e-3) / 1e12,
                               "fp64_frac": fl / (kms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, "neg_loglik": -v, "launch": ts.ctx.last_launch_info()}
    model = tracking._objective_model(P(vals), ts, DT, CELL, None, 4, 1, 5, 1)
    t0 = time.perf_counter()
    pr = ts.predict(model)
    wall = time.perf_counter() - t0
    kms = ts.ctx.last_kernel_ms()
    obytes = pr[0].nbytes
    out["c5_predict_Bs_F5"] = {"what": "configs[4] annotation: predict_Bs posteriors [5e5, 60, 4] (nb_substeps 1, frame_len 5), incl. the copy to the host",
                               "tracks_per_s": N5 / wall, "seconds": wall, "kernel_ms": kms, "kernel_tracks_per_s": N5 / (kms * 1e-3),
                               "algorithmic_bytes": nbytes + obytes, "hbm_gbs": (nbytes + obytes) / (kms * 1e-3) / 1e9,
                               "rowsum_err": float(np.abs(pr[0].sum(-1) - 1).max()), "launch": ts.ctx.last_launch_info()}
    del pr
    # the same model (4 states x 3 substeps) through the kernel v1.6.3's param_fitting calls: 4^4 x 4^3 = 16 384 expanded sequences at the
    # second position - beyond the 8192 whose plan arrays fit the LDS, refused until round 4
    try:
        import math
        Tsub = 1 - np.exp(-(Tm - np.diag(np.diag(Tm))) / 3)
        Tsub[np.arange(4), np.arange(4)] = 0
        Tsub[np.arange(4), np.arange(4)] = 1 - Tsub.sum(1)
        dsb = np.sqrt(2 * np.maximum(np.array([0.0, 0.02, 0.1, 0.5]), 1e-4) * DT)
        m16 = ts.make_model(np.array([[[LOCERR]]]), dsb, np.array([.25] * 4), Tsub, 0.1, tuple(CELL), 3, 4)
        w16, k16, v16 = timed(lambda: ts.loglik_th(m16, 0.2, 120, 2000), 2, warm=1)
        out["c5_loglik_threshold_ns3"] = {"what": "configs[4] model (4 states, nb_substeps 3, frame_len 4) through the threshold-fusion kernels: 16 384 expanded sequences "
                                                  "at the second position, the plan kernel's per-step arrays in its global workspace, the apply kernel reads that step's member list from global memory",
                                          "ms_per_eval": w16 * 1e3, "kernel_ms": k16, "tracks_per_s": N5 / w16, "finite": bool(math.isfinite(v16)),
                                          "launch": ts.ctx.last_launch_info()}
    except Exception as e:  # noqa: BLE001
        out["c5_loglik_threshold_ns3"] = {"error": str(e)}
    ts.close()
    # CPU side of the same configs (plain-C restatement, OpenMP, bounded samples)
    try:
        if no_cpu_baseline:
            raise RuntimeError("skipped (--no-cpu-baseline)")
        from oracle import oracle_c, oracle_np as O
        cores = _one_socket_cores()
        LocErr, ds, Fs, T, pBL = O.extract_params(vals, DT, nb_substeps=3, Matrix_type=1)  # the ORACLE's signature (values, dt, nb_substeps, Matrix_type)
        smp = synth.brownian_tracks(2 * cores, L5, [0.0, 0.02, 0.1, 0.5], Tm, [0.25] * 4, seed=3)
        t0 = time.perf_counter()
        oracle_c.run(smp, LocErr, ds, Fs, T, pBL, 0, O.p_stay_table(ds, 4, 3, CELL), 3, 4, L5, nthreads=cores)
        dtc = time.perf_counter() - t0
        out["c5_loglik_ns3_F4"]["cpu_port_c"] = {"tracks_per_s": len(smp) / dtc, "cores": cores, "sample": "%d tracks" % len(smp)}
    except Exception as e:
        out["c5_loglik_ns3_F4"]["cpu_port_c"] = {"error": str(e)}
    # ---- models the LDS kernels used to refuse (round 4, csrc/xt_big.h: one lane per track, sequence state in global memory)
    try:
        big = {}
        for S_, F_, nb_, Lb in ((5, 6, 20000, 20), (2, 12, 100000, 30)):
            Tmb = np.full((S_, S_), 0.05 / (S_ - 1))
            Tmb[np.arange(S_), np.arange(S_)] = 0.95
            Dsb = list(np.linspace(0.0, 0.5, S_))
            Csb = synth.brownian_tracks(nb_, Lb, Dsb, Tmb, [1.0 / S_] * S_, seed=50 + S_)
            vb = dict(LocErr=0.02, pBL=0.1)
            for i in range(S_):
                vb["D%d" % i], vb["F%d" % i] = max(Dsb[i], 1e-4), 1.0 / S_
                for j in range(S_):
                    if i != j:
                        vb["p%d%d" % (i, j)] = 0.05 / (S_ - 1)
            ts = tracking.TrackSet([Csb], device=device)
            mb = tracking._objective_model(P(vb), ts, DT, CELL, None, S_, 1, F_, 1)
            wall, kms, v = timed(lambda: ts.loglik(mb), 2)
            big["%d_states_F%d" % (S_, F_)] = {"tracks": nb_, "len": Lb, "sequences_per_track": S_ ** F_, "ms_per_eval": wall * 1e3, "kernel_ms": kms,
                                               "tracks_per_s": nb_ / (kms * 1e-3), "state_bytes_streamed_per_eval": 2.0 * nb_ * (Lb - 2) * (S_ ** F_) * 36,
                                               "neg_loglik": -v, "launch": ts.ctx.last_launch_info()}
            ts.close()
            del Csb
        out["big_models_global_state"] = dict(big, what="fixed-window likelihood of models whose sequence state does not fit a workgroup (5 states at the reference's "
                                              "default frame_len 6: 15 625 sequences per track; 2 states at frame_len 12: 2 048 groups): refused until round 3, now one "
                                              "lane per track with the state streamed through global memory")
    except Exception as e:  # noqa: BLE001
        out["big_models_global_state"] = {"error": str(e)}
    # ---- configs[0]-size dataset (a real experiment: ~7 000 tracks in 16 length buckets): one evaluation of both objectives and a whole fit
    Ds2, Tm2, Fs2 = [0.0, 0.25], [[0.9, 0.1], [0.1, 0.9]], [0.6, 0.4]
    sizes = synth.bucket_sizes_geometric(6730, list(range(5, 21)), 0.85)
    small = {str(L): synth.brownian_tracks(n, L, Ds2, Tm2, Fs2, seed=L) for L, n in sizes.items() if n > 0}
    _, lst, _ = tracking.engine.sort_buckets(small)
    ts = tracking.TrackSet(lst, device=device)
    v2 = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    model = tracking._objective_model(P(v2), ts, DT, CELL, None, 2, 1, 6, 1)
    w_win, k_win, _ = timed(lambda: ts.loglik(model), 50, warm=2)
    w_th, k_th, _ = timed(lambda: ts.loglik_th(model, 0.2, 120, 2000), 50, warm=2)
    ts.th_freeze_plan(True)
    w_thz, k_thz, _ = timed(lambda: ts.loglik_th(model, 0.2, 120, 2000), 50, warm=2)
    ts.th_freeze_plan(False)
    ts.close()
    fits = {}
    for grad in (() if no_fits else ("analytic", "fd", None)):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            r = tracking.param_fitting(small, DT, nb_states=2, frame_len=6, cell_dims=CELL, verbose=0, gradient=grad, device=device)
            fits["default" if grad is None else grad] = {"seconds": time.perf_counter() - t0, "objective_calls": int(r.nfev),
                                                          "gradient_calls": int(getattr(r, "ngev", 0)), "neg_loglik": float(r.residual[0])}
    fits_th = {}
    for grad in (() if no_fits else ("analytic", "fd")):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            r = tracking.param_fitting(small, DT, nb_states=2, frame_len=6, cell_dims=CELL, verbose=0, gradient=grad, device=device, fusion="threshold")
            fits_th[grad] = {"seconds": time.perf_counter() - t0, "objective_calls": int(r.nfev), "gradient_calls": int(getattr(r, "ngev", 0)),
                             "neg_loglik": float(r.residual[0])}
    out["c1_like_small_dataset"] = {"what": "6 730 tracks in 16 length buckets (5-20 positions), 2 states, frame_len 6: one evaluation, and param_fitting from the "
                                            "default start with the exact gradient / with finite differences",
                                    "window_ms_per_eval": w_win * 1e3, "window_kernel_ms": k_win, "threshold_ms_per_eval": w_th * 1e3,
                                    "threshold_kernels_ms": k_th, "threshold_frozen_plan_ms_per_eval": w_thz * 1e3, "threshold_frozen_plan_kernel_ms": k_thz,
                                    "fit": fits, "fit_threshold_fusion": fits_th}
    # ---- state-duration histograms and position refinement on 1e5 tracks x 30
    from extrack_amd.histograms import len_hist
    from extrack_amd.refined_localization import position_refinement
    big = {"30": synth.brownian_tracks(100000, 30, Ds2, Tm2, Fs2, seed=1)}
    pp = P(v2)
    with contextlib.redirect_stdout(io.StringIO()):
        len_hist(big, pp, DT, cell_dims=CELL, nb_states=2, max_nb_states=500, device=device)
        t0 = time.perf_counter()
        h = len_hist(big, pp, DT, cell_dims=CELL, nb_states=2, max_nb_states=500, device=device)
        t_h = time.perf_counter() - t0
        dsr = np.sqrt(2 * np.array([1e-3, 0.25]) * DT)
        position_refinement(big, 0.02, dsr, np.array(Fs2), np.array(Tm2), frame_len=6, device=device)
        t0 = time.perf_counter()
        mu, sg = position_refinement(big, 0.02, dsr, np.array(Fs2), np.array(Tm2), frame_len=6, device=device)
        t_r = time.perf_counter() - t0
    nb30 = big["30"].nbytes
    out["len_hist_1e5x30"] = {"what": "histograms.len_hist, 1e5 tracks x 30, 2 states, max_nb_states 500 (the reference's default)", "seconds": t_h,
                              "tracks_per_s": 1e5 / t_h, "hist_sum": float(h.sum()),
                              "algorithmic_bytes": nb30 + h.nbytes, "byte_model": "algorithmic: one read of the tracks (480 B / track) + the [29, 2] histogram.  The <= 500 surviving "
                              "sequences of a track (parent arrays, ~25 KB per copy) live in a per-workgroup region of global memory at this max_nb_states "
                              "(that frees LDS for 5 workgroups per CU instead of 2: 78 -> 62 ms); the region is rewritten every position and stays in "
                              "L2 / Infinity Cache (1 280 workgroups x 50 KB), the counters see ~114 GB of L2 <-> fabric traffic per 1e5 tracks "
                              "(profiles/r03_pmc_summary.txt)",
                              "hbm_gbs": (nb30 + h.nbytes) / t_h / 1e9}
    out["position_refinement_1e5x30"] = {"what": "refined_localization.position_refinement, 1e5 tracks x 30, 2 states, frame_len 6, threshold 0.1",
                                         "seconds": t_r, "tracks_per_s": 1e5 / t_r, "mean_sigma": float(sg["30"].mean()),
                                         "algorithmic_bytes": 2 * nb30 + sg["30"].nbytes,
                                         "byte_model": "tracks in (480 B / track) + refined positions and stds out (720 B / track); the per-position records of the "
                                         "forward and backward pass (~24 sequences x 4 doubles per position and pass) go through HBM once each way on top of that",
                                         "hbm_gbs": (2 * nb30 + sg["30"].nbytes) / t_r / 1e9}
    if no_cpu_baseline:  # e.g. under rocprofv3: no fork pools, no CPU legs
        return out
    try:  # CPU side: the numpy restatements (pinned to the reference's fixtures) on bounded samples, one socket's cores
        import multiprocessing as mp
        cores = _one_socket_cores()
        nh = 50 * cores
        smp = {"30": big["30"][:nh]}
        chunks = [({"30": smp["30"][a_:a_ + 50]}, v2) for a_ in range(0, nh, 50)]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(cpu_hist_worker, chunks)
        out["len_hist_1e5x30"]["cpu_baseline"] = {"tracks_per_s": nh / (time.perf_counter() - t0), "cores": cores, "kind": "port",
                                                  "sample": "%d tracks in 50-track chunks (len_hist's chunk), numpy oracle_hist, max_nb_states 500" % nh}
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(cpu_refine_worker, [({"30": smp["30"][a_:a_ + 50]}, dsr) for a_ in range(0, nh, 50)])
        out["position_refinement_1e5x30"]["cpu_baseline"] = {"tracks_per_s": nh / (time.perf_counter() - t0), "cores": cores, "kind": "port",
                                                            "sample": "%d tracks in 50-track buckets, numpy oracle_refine" % nh}
    except Exception as e:
        out["len_hist_1e5x30"]["cpu_baseline"] = {"error": str(e)}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set), relay rank 0's JSON line as the LAST stdout line and
    return the first non-zero exit code.  Runs before anything in this process has touched torch or the GPU; the children are fresh
    interpreters (Popen of a new program from a process that never initialised HIP), never an exec of a GPU-initialised process."""
    import subprocess
    import threading
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0's stdout is parsed here; the other ranks print nothing on stdout by contract, whatever they do print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    result = []

    def pump():
        for line in procs[0].stdout:
            t = line.strip()
            if t.startswith("{") and t.endswith("}") and '"metric"' in t:
                result.append(t)       # held back: printed last
            else:
                sys.stdout.write(line)
                sys.stdout.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for o in alive:       # a rank died: its peers would wait in the next collective for ever
                    procs[o].terminate()
        time.sleep(0.05)
    th.join(10)
    if result:
        print(result[-1], flush=True)
    elif rc == 0:
        rc = 1
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c2", "c2s", "c4"), default=None,
                    help="c2: BASELINE configs[1], 1e6 tracks PER GPU (weak scaling); c2s: the same 1e6 tracks IN TOTAL, row-sharded (strong "
                         "scaling - what BASELINE.json's metric names); c4: configs[3], 1e7 tracks in total, row-sharded (strong).  Default: "
                         "c2 on one GPU; on N > 1 GPUs c2s is the headline and c2 (weak) + c4 (strong) are measured in the same run (`scaling_runs`)")
    ap.add_argument("--tracks", type=int, default=None, help="tracks per GPU (c2) / in total (c2s, c4); default = the BASELINE config")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2] / configs[4] measurements after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fits", action="store_true", help="extra block: single evaluations of every kernel only, no whole param_fitting runs (profiler passes)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend: nccl = RCCL over xGMI (the measurement); gloo = rehearsal of the multi-rank control flow on a box "
                         "without GPUs (needs --rehearsal-context; nothing it prints is a measurement)")
    ap.add_argument("--rehearsal-context", default=None, metavar="MODULE:CLASS",
                    help="gloo rehearsal only (tests/test_bench_gloo.py): class under tests/ that stands in for the device context")
    argv = list(sys.argv[1:] if argv is None else argv)
    a = ap.parse_args(argv)
    if a.backend == "nccl" and a.rehearsal_context:
        raise SystemExit("--rehearsal-context is only valid with --backend gloo: the measurement always runs the HIP library")
    if a.backend == "gloo" and not a.rehearsal_context:
        raise SystemExit("--backend gloo is a CPU rehearsal of the control flow and needs --rehearsal-context (there is no CPU fallback)")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(_launch_ranks(a.gpus, argv))

    import torch
    from extrack_amd import synth, tracking
    from extrack_amd.lmfit_compat import Parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or plainly: python bench.py --gpus N)" % (a.gpus, world))
    on_gpu = a.backend == "nccl"
    if not on_gpu:
        import importlib
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        mod, cls = a.rehearsal_context.split(":")
        from extrack_amd import _lib as _xl
        _xl.Context = getattr(importlib.import_module(mod), cls)
    if on_gpu:
        torch.cuda.set_device(local)
    comm = None
    rccl = None
    if world > 1 or os.environ.get("EXTRACK_BENCH_FORCE_COMM") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: the bootstrap never needs an external interface
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        from extrack_amd.distributed import Comm
        comm = Comm()
        names = [None] * world
        dist.all_gather_object(names, "cuda:%d %s" % (local, torch.cuda.get_device_name(local)) if on_gpu else "cpu (rehearsal)")
        rccl = {"backend": dist.get_backend(), "world": world, "devices": names,
                "collective": "one all-reduce(sum) of {sum LL, failure flag} (2 doubles) per evaluation, on the kernels' stream"}

    p = Parameters()
    for k, v in dict(D0=DS_COEF[0], D1=DS_COEF[1], LocErr=LOCERR, F0=FS[0], F1=FS[1], p01=0.1, p10=0.1, pBL=PBL).items():
        p.add(k, value=v)

    def barrier():
        if comm is not None:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def run_config(cfg, tracks_arg, keep=False):
        """One workload: this rank's rows resident in HBM, `warmup` untimed steps, then exactly `steps` steps between two
        barrier + synchronize brackets; the time is the MAX over the ranks."""
        from extrack_amd.distributed import shard_plan
        if cfg == "c2":
            n_loc = tracks_arg if tracks_arg else N_TRACKS
            total, scaling = n_loc * world, "weak"
        else:
            total = tracks_arg if tracks_arg else (N_TRACKS if cfg == "c2s" else 10 * N_TRACKS)
            lo_, hi_ = shard_plan([total], [LEN], world)[0][rank]   # this rank's rows of the one bucket
            n_loc, scaling = hi_ - lo_, "strong"
        Cs = synth.brownian_tracks(n_loc, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=rank)
        ts = tracking.TrackSet([Cs], device=local, min_len=LEN, max_len=LEN, allow_empty=True)
        del Cs
        model = tracking._objective_model(p, ts, DT, CELL, None, S, NS, FRAME, 1)
        step = (lambda: ts.loglik(model)) if comm is None else (lambda: comm.allreduce_loglik(ts, model))
        # untimed set-up before the contract's W warm-up steps: the GPU clock needs ~20 launches of a fresh process to settle (the first launches run
        # 25 % slower, profiles/r04_headline_timed_launches.txt) - a fixed count (every rank makes the same collective calls), reported in the line
        for _ in range(SETTLE_LAUNCHES if on_gpu else 0):
            step()
        for _ in range(a.warmup):
            val = step()
        kernel_ms = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            val = step()
            kernel_ms.append(ts.ctx.last_kernel_ms() if n_loc else 0.0)
        barrier()
        dt_loc = time.perf_counter() - t0
        dt_all = comm.allreduce_scalar(dt_loc, "max") if comm is not None else dt_loc
        k_ms = float(np.mean(kernel_ms))
        k_max = comm.allreduce_scalar(k_ms, "max") if comm is not None else k_ms
        res = {"config": cfg, "scaling": scaling, "total_tracks": total, "tracks_per_gpu": n_loc, "ms_per_step": dt_all / a.steps * 1e3,
               "kernel_ms": k_ms, "kernel_ms_max_over_ranks": k_max, "comm_ms": dt_all / a.steps * 1e3 - k_max,
               "value": (total / N_TRACKS) * a.steps / dt_all, "unit": "1e6-track LL evals/s", "neg_loglik": -val,
               "launch": ts.ctx.last_launch_info() if n_loc else {}, "clock_settle_launches": SETTLE_LAUNCHES if on_gpu else 0}
        if keep:
            return res, ts, model
        ts.close()
        return res, None, None

    # the headline workload (resident in HBM before the timed region); N > 1 without --config: all three scaling workloads in this run
    primary = a.config if a.config else ("c2" if world == 1 else "c2s")
    scaling_runs = {}
    if a.config is None and world > 1:
        for cfg in ("c2", "c4"):
            scaling_runs[cfg] = run_config(cfg, None)[0]
    head, ts, model = run_config(primary, a.tracks, keep=True)
    scaling_runs[primary] = dict(head)
    a.tracks, total_tracks, scaling = head["tracks_per_gpu"], head["total_tracks"], head["scaling"]
    dt_all = head["ms_per_step"] * 1e-3 * a.steps
    kernel_ms = [head["kernel_ms"]]
    val = -head["neg_loglik"]
    a.config = primary
    # secondary measurement (outside the timed region, one GPU only): the threshold-fusion kernel that the reference's
    # current param_fitting calls (tracking.py:427-743), same data, v1.6.3 defaults (threshold 0.2, max_nb_states 120,
    # 2000-track chunks): plan kernel + apply kernel per evaluation
    launch_info = ts.ctx.last_launch_info()
    th = None
    grad_info = None
    if on_gpu and world == 1 and a.tracks == N_TRACKS and a.config == "c2":
        from extrack_amd import gradient
        pg = tracking.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, DS_COEF[1]], estimated_LocErr=[LOCERR], estimated_Fs=[FS[0]],
                                      estimated_transition_rates=0.1)
        gnames = gradient.free_names(pg)
        for _ in range(2):
            gv, gg = gradient.objective_and_gradient(pg, ts, DT, CELL, S, NS, FRAME, names=gnames)
        grad_info = {"what": "BASELINE configs[1] data: -sum(LL) AND its exact gradient (7 free parameters) in one pass of the gradient kernel "
                             "(extrack_loglik_grad), the evaluation an analytic-gradient BFGS iteration costs",
                     "kernel_ms": ts.ctx.last_grad_ms(), "n_directions": len(gnames), "grad_inf_norm": float(np.abs(gg).max())}
    if on_gpu and world == 1 and a.tracks == N_TRACKS and a.config == "c2":
        for _ in range(2):
            th_val = ts.loglik_th(model, 0.2, 120, 2000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        th_ms = []
        for _ in range(max(3, a.steps // 2)):
            th_val = ts.loglik_th(model, 0.2, 120, 2000)
            th_ms.append(ts.ctx.last_kernel_ms())
        th_dt = (time.perf_counter() - t1) / max(3, a.steps // 2)
        ts.th_freeze_plan(True)   # the plan of the last evaluation kept: what the optimiser sees between two re-plannings (apply kernel only)
        for _ in range(2):
            ts.loglik_th(model, 0.2, 120, 2000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(3, a.steps // 2)):
            fz_val = ts.loglik_th(model, 0.2, 120, 2000)
        fz_dt = (time.perf_counter() - t1) / max(3, a.steps // 2)
        fz_k = ts.ctx.last_kernel_ms()
        ts.th_freeze_plan(False)
        th = {"frozen_plan": {"what": "the same evaluation with the merge plan of the previous one kept (extrack_th_freeze_plan): no plan kernel, no read-back - every "
                                      "evaluation between two re-plannings of a frozen-plan fit", "ms_per_eval": fz_dt * 1e3, "kernel_ms": fz_k,
                              "same_value": bool(fz_val == th_val)},
              "what": "P_Cs_inter_bound_stats_th path (threshold 0.2, max_nb_states 120, chunk 2000): plan + apply kernels",
              "value": 1.0 / th_dt, "unit": "1e6-track LL evals/s", "ms_per_eval": th_dt * 1e3, "kernels_ms": float(np.mean(th_ms)),
              "hbm_gbs": a.tracks * LEN * DIMS * 8 / (float(np.mean(th_ms)) * 1e-3) / 1e9, "neg_loglik": -th_val,
              "launch": ts.ctx.last_launch_info()}
    ts.close()
    extra = None
    if on_gpu and world == 1 and a.config == "c2" and a.tracks == N_TRACKS and not a.no_extra:
        extra = other_configs(local, a.no_cpu_baseline, a.no_fits)
        # configs[3] on ONE GPU: the N = 1 point of the strong-scaling curve `scaling_runs.c4` reports at N > 1
        scaling_runs["c4"] = run_config("c4", None)[0]
    if comm is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = dt_all / a.steps * 1e3
    evals_per_s = (total_tracks / N_TRACKS) * a.steps / dt_all
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = a.tracks * LEN * DIMS * 8          # one read of the track, LL reduced in-kernel
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0  # no device timer in the gloo rehearsal
    traffic, traffic_round = None, None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json)
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and a.tracks == N_TRACKS:
        try:
            tj = json.load(open(tpath))
            traffic, traffic_round = tj.get("bytes_per_launch"), tj.get("round")
        except Exception:
            traffic = None
    # secondary (honest) bound: fp64 vector issue, from the instruction mix of the steady-state step of the CURRENT build
    # (tools/isa_mix.py -> profiles/isa_mix.json; regenerated whenever csrc/xt_reg2.h changes)
    mix = None
    mpath = os.path.join(ROOT, "profiles", "isa_mix.json")
    if os.path.exists(mpath):
        try:
            mix = json.load(open(mpath))
        except Exception:
            mix = None
    out = {
        "metric": "log-likelihood evals/sec (1e6 tracks, 2-state, len=30)", "value": evals_per_s, "unit": "1e6-track LL evals/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic" if on_gpu else "synthetic (gloo REHEARSAL, not a measurement)",
        "config": {"workload": ("BASELINE configs[1]: %d tracks/GPU, 2 states, len=30, 2-D, nb_substeps=1, frame_len=6, "
                                "single log-likelihood eval per step" % a.tracks) if a.config == "c2" else
                               ("BASELINE configs[%d]: %d tracks in total, row-sharded over %d GPU(s) (%d on rank 0), 2 states, len=30, 2-D, "
                                "nb_substeps=1, frame_len=6, single log-likelihood eval per step (local kernel + one RCCL all-reduce)"
                                % (1 if a.config == "c2s" else 3, total_tracks, world, a.tracks)),
                   "name": a.config, "tracks_per_gpu": a.tracks, "total_tracks": total_tracks, "parallelism": "dp%d" % world, "launch": launch_info,
                   "clock_settle_launches": head["clock_settle_launches"],
                   "clock_settle_note": "untimed launches of the same step before the W warm-up steps: a fresh process reaches its sustained GPU clock after ~20 launches"},
        "kernel_ms": head["kernel_ms_max_over_ranks"], "comm_ms": head["comm_ms"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_round": traffic_round,
                     "traffic_source": "static: rocprofv3 PMC passes of an earlier run of this workload, profiles/hbm_traffic.json "
                                       "(not re-measured inside this run; `traffic_round` = the round it was collected in)",
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "the recursion is FP64-VALU bound, not HBM bound (arithmetic intensity ~300 flop/B, DESIGN.md)",
                     },
        "neg_loglik": -val,
    }
    if mix is not None and k_ms > 0:
        wave_steps = a.tracks / mix["tracks_per_wave"] * (LEN - 1)
        flop_per_eval = wave_steps * mix["flop_per_wave_step"]
        tflops = flop_per_eval / (k_ms * 1e-3) / 1e12
        out["roofline"]["fp64_valu"] = {
            "achieved": tflops, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TF, "flop_per_launch": flop_per_eval,
            "valu_issue_floor_ms": wave_steps * mix["issue_cycles_per_wave_step"] / (256 * 4) / 2.4e9 * 1e3,
            "instructions_per_wave_step": {"fp64": mix["fp64_valu_per_wave_step"], "valu32": mix["valu32_per_wave_step"], "lds": mix["lds_per_wave_step"]},
            "source": "profiles/isa_mix.json (tools/isa_mix.py on the current sources): " + mix["kernel"]}
    if rccl is not None:
        out["rccl"] = rccl
    out["scaling_runs"] = dict(scaling_runs, note="every workload of this run: c2 = 1e6 tracks per GPU (weak), c2s = 1e6 tracks in total (strong; the "
                               "headline at N > 1), c4 = 1e7 tracks in total (strong); comm_ms = ms_per_step - kernel_ms (max over ranks): launch, "
                               "all-reduce and read-back")
    if th is not None:
        out["threshold_fusion"] = th
    if grad_info is not None:
        grad_info["fd_equivalent_ms"] = (grad_info["n_directions"] + 1) * k_ms
        out["loglik_gradient"] = grad_info
    if extra is not None:
        out["extra"] = extra
    if not a.no_cpu_baseline and world == 1 and on_gpu:
        out["cpu_baseline"] = cpu_baseline()
        out["speedup_vs_cpu_baseline"] = evals_per_s / out["cpu_baseline"]["value"]
        try:
            out["cpu_baseline_compiled"] = cpu_baseline_compiled(out["cpu_baseline"]["cores"])
        except Exception as e:  # gcc missing on the box: the numpy baseline stands alone
            out["cpu_baseline_compiled"] = {"error": str(e)}
        if th is not None:
            th["cpu_baseline"] = cpu_baseline_th(out["cpu_baseline"]["cores"])
    try:  # RCCL's version banner (NCCL_DEBUG=VERSION) sits in the C stdio buffer: flush it so the JSON line comes last
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
