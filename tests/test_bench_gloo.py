"""Rehearsal of bench.py's multi-rank control flow on CPU: ``python bench.py --gpus 2 --backend gloo ...`` is invoked PLAINLY (no
RANK / WORLD_SIZE in the environment, no launcher) - bench.py starts its own two rank processes, the ranks run the weak (c2) and the
two strong (c2s, c4) workloads over a gloo group, and the parent relays rank 0's JSON line as the last stdout line.  The device context
is replaced by the oracle-backed stand-in of tests/test_distributed_gloo.py (``--rehearsal-context``), so that the self-launch / rank /
seed / shard / barrier / max-over-ranks / JSON-on-rank-0 logic has executed at world > 1 before the driver's 8-GPU job does.  Nothing
it prints is a measurement."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_distributed_gloo import OracleContext  # noqa: E402


class BenchContext(OracleContext):
    """The timing / launch queries of the real context (bench.py reads them every step)."""

    def last_kernel_ms(self):
        return 0.0

    def last_launch_info(self):
        return {}


def _bench(*argv, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--rehearsal-context",
                        "test_bench_gloo:BenchContext"] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    return r


def test_bench_self_launch_weak_and_strong():
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--tracks", "48", "--config", "c2")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().split("\n")[-1])   # the JSON line is the LAST stdout line
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["total_tracks"] == 96 and d["config"]["tracks_per_gpu"] == 48 and d["config"]["parallelism"] == "dp2"
    assert d["rccl"]["world"] == 2 and len(d["rccl"]["devices"]) == 2 and d["rccl"]["backend"] == "gloo"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and np.isfinite(d["neg_loglik"])
    assert abs(d["value"] - 96 / 1e6 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-6 * d["value"]  # whole-job units / max-over-ranks time
    assert "comm_ms" in d and "kernel_ms" in d
    r4 = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--tracks", "101", "--config", "c4")
    assert r4.returncode == 0, r4.stderr[-2000:]
    d4 = json.loads(r4.stdout.strip().split("\n")[-1])
    assert d4["scaling"] == "strong" and d4["config"]["total_tracks"] == 101 and d4["config"]["tracks_per_gpu"] in (50, 51)
    # every rank draws its own rows (seed = rank): the reduced value is the sum over both ranks' tracks, finite and reproducible
    r5 = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--tracks", "101", "--config", "c4")
    assert json.loads(r5.stdout.strip().split("\n")[-1])["neg_loglik"] == d4["neg_loglik"]


def test_bench_default_multi_gpu_run_reports_all_three_workloads():
    """No --config at N > 1: c2s (strong, 1e6 tracks in total at full size) is the headline, c2 (weak) and c4 (strong) ride along.
    EXTRACK_BENCH_N_TRACKS scales the "1e6" of all three down to what the oracle-backed stand-in evaluates in seconds."""
    os.environ["EXTRACK_BENCH_N_TRACKS"] = "60"
    try:
        r = _bench("--gpus", "2", "--steps", "1", "--warmup", "1")
    finally:
        os.environ.pop("EXTRACK_BENCH_N_TRACKS")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["scaling"] == "strong" and d["config"]["name"] == "c2s" and d["config"]["total_tracks"] == 60 and d["n_gpus"] == 2
    runs = d["scaling_runs"]
    assert runs["c2"]["scaling"] == "weak" and runs["c2"]["total_tracks"] == 120
    assert runs["c4"]["scaling"] == "strong" and runs["c4"]["total_tracks"] == 600
    assert runs["c2s"]["total_tracks"] == 60 and runs["c2s"]["tracks_per_gpu"] == 30
    for k in ("c2", "c2s", "c4"):
        assert runs[k]["ms_per_step"] > 0 and "comm_ms" in runs[k] and np.isfinite(runs[k]["neg_loglik"])


def test_bench_propagates_a_failing_rank():
    """A rank that dies (here: both fail to find their context class) ends the run with a non-zero exit code and no JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--rehearsal-context", "test_bench_gloo:Missing",
                        "--gpus", "2", "--steps", "1", "--warmup", "0", "--tracks", "8", "--config", "c2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout
