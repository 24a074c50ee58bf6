#!/usr/bin/env python3
"""Headline benchmark: log-likelihood evaluations per second on synthetic track batches.

Workload at N GPUs (weak scaling): every rank holds BASELINE.json configs[1] - 1e6 tracks, 2 states, length 30,
2-D, nb_substeps=1, frame_len=6 - resident in HBM; a "step" is one evaluation of -sum(LL) over all ranks' tracks
(local kernel + one all-reduce of the fp64 scalar over RCCL).  `value` = (N x 1e6-track evaluations) / s, i.e. in
units of "1e6-track log-likelihood evaluations per second".

    python bench.py --gpus 1 --steps 20 --warmup 3
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

N_TRACKS, LEN, DIMS, S, NS, FRAME = 1000000, 30, 2, 2, 1, 6
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tracks", type=int, default=N_TRACKS, help="tracks per GPU (default = the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from extrack_amd import synth, tracking
    from extrack_amd.lmfit_compat import Parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    torch.cuda.set_device(local)
    comm = None
    if world > 1 or os.environ.get("EXTRACK_BENCH_FORCE_COMM") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: the bootstrap never needs an external interface
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        from extrack_amd.distributed import Comm
        comm = Comm()

    # synthetic data of the BASELINE shape, resident in HBM before the timed region
    Cs = synth.brownian_tracks(a.tracks, LEN, DS_COEF, TRMAT, FS, LOCERR, DT, DIMS, seed=rank)
    ts = tracking.TrackSet([Cs], device=local, min_len=LEN, max_len=LEN)
    del Cs
    p = Parameters()
    for k, v in dict(D0=DS_COEF[0], D1=DS_COEF[1], LocErr=LOCERR, F0=FS[0], F1=FS[1], p01=0.1, p10=0.1, pBL=PBL).items():
        p.add(k, value=v)
    model = tracking._objective_model(p, ts, DT, CELL, None, S, NS, FRAME, 1)

    def step():
        return ts.loglik(model) if comm is None else comm.allreduce_loglik(ts, model)

    def barrier():
        if comm is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        val = step()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        val = step()
        kernel_ms.append(ts.ctx.last_kernel_ms())
    barrier()
    dt_loc = time.perf_counter() - t0
    if comm is not None:
        dt_all = comm.allreduce_scalar(dt_loc, "max")
    else:
        dt_all = dt_loc
    # secondary measurement (outside the timed region, one GPU only): the threshold-fusion kernel that the reference's
    # current param_fitting calls (tracking.py:427-743), same data, v1.6.3 defaults (threshold 0.2, max_nb_states 120,
    # 2000-track chunks): plan kernel + apply kernel per evaluation
    launch_info = ts.ctx.last_launch_info()
    th = None
    if world == 1 and a.tracks == N_TRACKS:
        for _ in range(2):
            th_val = ts.loglik_th(model, 0.2, 120, 2000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        th_ms = []
        for _ in range(max(3, a.steps // 2)):
            th_val = ts.loglik_th(model, 0.2, 120, 2000)
            th_ms.append(ts.ctx.last_kernel_ms())
        th_dt = (time.perf_counter() - t1) / max(3, a.steps // 2)
        th = {"what": "P_Cs_inter_bound_stats_th path (threshold 0.2, max_nb_states 120, chunk 2000): plan + apply kernels",
              "value": 1.0 / th_dt, "unit": "1e6-track LL evals/s", "ms_per_eval": th_dt * 1e3, "kernels_ms": float(np.mean(th_ms)),
              "hbm_gbs": a.tracks * LEN * DIMS * 8 / (float(np.mean(th_ms)) * 1e-3) / 1e9, "neg_loglik": -th_val,
              "launch": ts.ctx.last_launch_info()}
    if comm is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = dt_all / a.steps * 1e3
    evals_per_s = world * (a.tracks / N_TRACKS) * a.steps / dt_all
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = a.tracks * LEN * DIMS * 8          # one read of the track, LL reduced in-kernel
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/gpu_pmc.sh -> profiles/hbm_traffic.json)
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and a.tracks == N_TRACKS:
        try:
            traffic = json.load(open(tpath)).get("bytes_per_launch")
        except Exception:
            traffic = None
    # secondary (honest) bound: fp64 vector issue.  Flop count per track-step from the kernel's ISA:
    # 34 FMA + 50 other fp64 instructions per wave-step of 2 tracks -> 118 flop x 64 lanes / 2 tracks = 3776 flop.
    flop_per_eval = a.tracks * (LEN - 1) * 3776.0
    valu_issue_cycles = a.tracks / 2 * (LEN - 1) * (84 * 4 + 37 * 2)   # fp64 ops issue in 4 cycles/wave, 32-bit ops in >= 2
    tflops = flop_per_eval / (k_ms * 1e-3) / 1e12
    out = {
        "metric": "log-likelihood evals/sec (1e6 tracks, 2-state, len=30)", "value": evals_per_s, "unit": "1e6-track LL evals/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d tracks/GPU, 2 states, len=30, 2-D, nb_substeps=1, frame_len=6, "
                               "single log-likelihood eval per step" % a.tracks,
                   "tracks_per_gpu": a.tracks, "parallelism": "dp%d" % world, "launch": launch_info},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "the recursion is FP64-VALU bound, not HBM bound (arithmetic intensity ~300 flop/B, DESIGN.md)",
                     "fp64_valu": {"achieved": tflops, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TF,
                                   "flop_per_launch": flop_per_eval,
                                   "valu_issue_floor_ms": valu_issue_cycles / (256 * 4) / 2.4e9 * 1e3}},
        "neg_loglik": -val,
    }
    if th is not None:
        out["threshold_fusion"] = th
    if not a.no_cpu_baseline and world == 1:
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
