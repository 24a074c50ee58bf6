"""Per-evaluation cost of going through the communicator (1 rank, RCCL) on the GPU box: fixed-window and threshold-fusion objectives,
C2-size (1e6 x 30) and a small real-world-size dataset.  python tools/gpu_comm_overhead.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as dist
from extrack_amd import synth, tracking as T
from extrack_amd.distributed import Comm

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
comm = Comm()
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
p = T.Parameters()
for k, v in vals.items():
    p.add(k, value=v)


def timeit(f, n):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, lens, n in (("C2 1e6 x 30", {30: 1_000_000}, 20), ("small: 6 730 tracks, 16 buckets", synth.bucket_sizes_geometric(6730, list(range(5, 21)), 0.85), 100)):
    lst = [synth.brownian_tracks(k, L, Ds, Tm, Fs, seed=L) for L, k in lens.items() if k > 0]
    for fusion, chunk in (("window", None), ("threshold", 2000)):
        ts = comm.shard_trackset(lst, chunk=chunk)
        kw = dict(verbose=0, fusion=fusion)
        a = timeit(lambda: T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, comm=comm, **kw), n)
        b = timeit(lambda: T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, **kw), n)
        print("%-34s %-9s with communicator %.3f ms   without %.3f ms   overhead %+.1f us" % (name, fusion, a, b, (a - b) * 1e3), flush=True)
        ts.close()
dist.destroy_process_group()
