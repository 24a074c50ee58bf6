"""Per-evaluation cost of everything that is not the kernel (1 rank, RCCL) on the GPU box: the strong-scaled shard of the headline
dataset at 8 GPUs (125 000 tracks), at 2 GPUs (500 000), the whole 1e6 and a small real-world-size dataset; fixed-window and
threshold-fusion objectives.  "step" = wall time per evaluation through the communicator (enqueue, all-reduce on the stream, pinned
read-back), "kernel" = HIP-event time of the kernels, "host" = step - kernel.   python tools/gpu_comm_overhead.py [out.txt]"""
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
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = Comm()
Ds, Tm, Fs = [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4]
vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
p = T.Parameters()
for k, v in vals.items():
    p.add(k, value=v)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def timeit(f, ts, n):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ks = []
    t0 = time.perf_counter()
    for _ in range(n):
        f()
        ks.append(ts.ctx.last_kernel_ms())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(np.mean(ks))


say("%-36s %-9s %10s %10s %10s %12s" % ("dataset", "fusion", "step ms", "kernel ms", "host us", "no-comm ms"))
sets = (("125 000 x 30 (1e6 / 8 GPUs)", {30: 125000}, 200), ("500 000 x 30 (1e6 / 2 GPUs)", {30: 500000}, 100), ("1e6 x 30", {30: 1000000}, 50),
        ("small: 6 730 tracks, 16 buckets", synth.bucket_sizes_geometric(6730, list(range(5, 21)), 0.85), 200))
for name, lens, n in sets:
    lst = [synth.brownian_tracks(k, L, Ds, Tm, Fs, seed=L) for L, k in lens.items() if k > 0]
    for fusion, chunk in (("window", None), ("threshold", 2000)):
        ts = comm.shard_trackset(lst, chunk=chunk)
        model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
        if fusion == "window":
            a, ka = timeit(lambda: comm.allreduce_loglik(ts, model), ts, n)
            b, kb = timeit(lambda: ts.loglik(model), ts, n)
        else:
            a, ka = timeit(lambda: comm.allreduce_loglik_th(ts, model, 0.2, 120, 2000), ts, n)
            b, kb = timeit(lambda: ts.loglik_th(model, 0.2, 120, 2000), ts, n)
        say("%-36s %-9s %10.4f %10.4f %10.1f %12.4f" % (name, fusion, a, ka, (a - ka) * 1e3, b))
        ts.close()
dist.destroy_process_group()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
