"""Rehearsal of bench.py's multi-rank control flow on CPU (verdict r2, item 8): two gloo ranks run ``bench.main`` for ``--config c2`` and
``--config c4`` with the device context replaced by the oracle-backed stand-in of tests/test_distributed_gloo.py, so that the rank /
seed / shard / barrier / max-over-ranks / JSON-on-rank-0 logic has executed at world > 1 before the driver's 8-GPU job does.  Nothing
it prints is a measurement."""
import json
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q, argv):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import contextlib
    import io
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from extrack_amd import _lib
    from test_distributed_gloo import OracleContext

    class Ctx(OracleContext):  # the timing / launch queries of the real context
        def last_kernel_ms(self):
            return 0.0

        def last_launch_info(self):
            return {}
    _lib.Context = Ctx
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(argv)
    q.put((rank, buf.getvalue()))


def _run(argv, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, argv)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        outs = dict(q.get(timeout=240) for _ in range(world))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return outs


def test_bench_multi_rank_branch_weak_and_strong():
    outs = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--tracks", "48", "--config", "c2"])
    assert outs[1].strip() == ""  # only rank 0 prints
    d = json.loads(outs[0].strip().split("\n")[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["total_tracks"] == 96 and d["config"]["tracks_per_gpu"] == 48 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and np.isfinite(d["neg_loglik"])
    assert abs(d["value"] - 96 / 1e6 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-6 * d["value"]  # whole-job units / max-over-ranks time
    outs = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--tracks", "101", "--config", "c4"])
    d4 = json.loads(outs[0].strip().split("\n")[-1])
    assert d4["scaling"] == "strong" and d4["config"]["total_tracks"] == 101 and d4["config"]["tracks_per_gpu"] in (50, 51)
    # the strong-scaling dataset is ONE seeded bucket cut by rows? no: every rank draws its own rows (seed = rank) - the reduced value is
    # the sum over both ranks' tracks and must be finite and reproducible
    outs2 = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--tracks", "101", "--config", "c4"])
    assert json.loads(outs2[0].strip().split("\n")[-1])["neg_loglik"] == d4["neg_loglik"]
