"""The N>1 path on CPU: world_size-2 gloo process group.  Sharding, the dataset-global min/max length agreement
and the scalar all-reduce of extrack_amd.distributed are exercised for real; the per-rank likelihood comes from the
oracle here (no GPU in this container) - on the GPU box the same Comm drives the HIP path over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from extrack_amd import synth
    from extrack_amd.distributed import Comm, shard_range
    from oracle import oracle_np as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        comm = Comm()
        assert comm.backend == "gloo" and comm.world == world
        vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
        lens = {5: 41, 9: 30, 12: 1}  # the longest bucket has ONE track: rank 1 owns none of it
        tracks = {str(L): synth.brownian_tracks(n, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=L) for L, n in lens.items()}
        lst = [tracks[k] for k in ("5", "9", "12")]
        t_loc, _ = comm.shard_buckets(lst)
        lo, hi = comm.global_min_max_len([b.shape[1] for b in t_loc])
        assert (lo, hi) == (5, 12)
        if rank == 1:
            assert 12 not in [b.shape[1] for b in t_loc]  # local max length differs from the global one
        # local objective with the GLOBAL min/max length (isBL of each bucket must not depend on the shard)
        LocErr, ds, Fs, T, pBL = O.extract_params(vals, 0.02, 1, 1)
        local = 0.0
        for b in t_loc:
            local += O.proba_cs(b, LocErr, ds, Fs, T, pBL, 0 if b.shape[1] == hi else 1, [1], 1, 6, lo).sum()
        total = comm.allreduce_scalar(local, "sum")
        ref = -O.cum_proba_cs(vals, tracks, 0.02, [1], None, 1, 6)
        covered = comm.allreduce_scalar(sum(len(b) for b in t_loc), "sum")
        # ordered gather of per-rank row blocks (what predict_Bs(comm=...) uses): rank 0 must get the rows in input order
        local = {k: v[slice(*shard_range(len(v), rank, world))] for k, v in tracks.items()}
        full = comm.gather_rows(local)
        if rank == 0:
            assert all(np.array_equal(full[k], tracks[k]) for k in tracks)
        else:
            assert full is None
        # threshold-fusion objective: shards are whole chunks, so every chunk keeps the pilot tracks it has on one GPU
        from oracle import oracle_th as OT
        chunk = 16
        th_loc = 0.0
        t_th, _ = comm.shard_buckets(lst, chunk=chunk)
        for b in t_th:
            for a0 in range(0, len(b), chunk):
                th_loc += OT.proba_cs_th(b[a0:a0 + chunk], LocErr, ds, Fs, T, pBL, 0 if b.shape[1] == hi else 1, [1], 1, 6, lo, 0.2, 120).sum()
        th_total = comm.allreduce_scalar(th_loc, "sum")
        th_ref = -OT.cum_proba_cs_th(vals, tracks, 0.02, [1], None, 1, 6, 1, 0.2, 120, chunk=chunk)
        assert abs(th_total - th_ref) < 1e-12 * abs(th_ref), (th_total, th_ref)
        assert [shard_range(41, r, world, chunk) for r in range(world)] == [(0, 32), (32, 41)]
        q.put((rank, total, ref, covered, [shard_range(41, r, world) for r in range(world)]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_objective_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, ref, covered, ranges in res:
        assert covered == 72
        assert abs(total - ref) < 1e-12 * abs(ref), (rank, total, ref)
        assert ranges == [(0, 21), (21, 41)]
    assert res[0][1] == res[1][1]  # every rank sees the same reduced scalar


def test_shard_range_partitions_exactly():
    from extrack_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
