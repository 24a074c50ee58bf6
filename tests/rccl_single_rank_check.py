"""Helper of tests/test_hip_parity.py::test_rccl_single_rank_allreduce_path (run as a subprocess on the GPU box)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

from extrack_amd import synth, tracking as T
from extrack_amd.distributed import Comm
from extrack_amd.lmfit_compat import Parameters

s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    comm = Comm()
    tr = {str(L): synth.brownian_tracks(300, L, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=L) for L in (6, 11)}
    _, lst, _ = T.engine.sort_buckets(tr)
    p = Parameters()
    for k, v in dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1).items():
        p.add(k, value=v)
    ts = comm.shard_trackset(lst)
    assert (ts.min_len, ts.max_len) == (6, 11)
    a = T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0, comm=comm)
    b = T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0)
    ts.close()
    assert a == b, (a, b)
    print("RCCL_PATH_OK", a)
finally:
    dist.destroy_process_group()
