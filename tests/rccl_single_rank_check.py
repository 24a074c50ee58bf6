"""Helper of tests/test_hip_parity.py::test_rccl_single_rank_allreduce_path (run as a subprocess on the GPU box)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
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
    # the same communicator through the public fitting entry point (sharding + per-evaluation all-reduce inside the optimiser loop)
    import contextlib, io
    p0 = T.generate_params(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, estimated_Ds=[0.0001, 0.1], estimated_Fs=[0.5],
                           estimated_transition_rates=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        # (gradient given explicitly: the default is decided by a timing probe, which two runs need not answer alike)
        fit_c = T.param_fitting(tr, 0.02, params=p0, nb_states=2, frame_len=4, verbose=0, method="bfgs", cell_dims=[1], comm=comm, gradient="fd")
        fit_s = T.param_fitting(tr, 0.02, params=p0, nb_states=2, frame_len=4, verbose=0, method="bfgs", cell_dims=[1], gradient="fd")
        fit_ca = T.param_fitting(tr, 0.02, params=p0, nb_states=2, frame_len=4, verbose=0, method="bfgs", cell_dims=[1], comm=comm, gradient="analytic")
        fit_sa = T.param_fitting(tr, 0.02, params=p0, nb_states=2, frame_len=4, verbose=0, method="bfgs", cell_dims=[1], gradient="analytic")
        fit_d = T.param_fitting(tr, 0.02, params=p0, nb_states=2, frame_len=4, verbose=0, method="bfgs", cell_dims=[1], comm=comm)  # probe + collective agreement
    assert fit_c.nfev == fit_s.nfev and fit_c.residual[0] == fit_s.residual[0], (fit_c.nfev, fit_s.nfev, fit_c.residual, fit_s.residual)
    # objective + gradient: the 1 + nvar doubles are written by the kernels into the buffer RCCL reduces (extrack_loglik_grad_async)
    assert fit_ca.nfev == fit_sa.nfev and fit_ca.residual[0] == fit_sa.residual[0] and fit_ca.ngev > 0, (fit_ca.nfev, fit_sa.nfev)
    assert abs(fit_d.residual[0] - fit_s.residual[0]) < 1e-6 * abs(fit_s.residual[0])
    from extrack_amd import gradient
    ts = comm.shard_trackset(lst)
    names = gradient.free_names(p0)
    v1, g1 = gradient.objective_and_gradient(p0, ts, 0.02, [1], 2, 1, 6, comm=comm, names=names)
    v2, g2 = gradient.objective_and_gradient(p0, ts, 0.02, [1], 2, 1, 6, names=names)
    assert v1 == v2 and np.array_equal(g1, g2), (v1, v2, g1, g2)
    # a failure while enqueueing this rank's kernels travels through the collective as a flag and is raised after it; the next evaluation works
    real = ts.ctx.loglik_async
    def boom(*a_, **k_):
        raise RuntimeError("injected enqueue failure")
    ts.ctx.loglik_async = boom
    try:
        T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0, comm=comm)
        raise SystemExit("the injected failure was swallowed")
    except RuntimeError as e:
        assert "injected" in str(e)
    ts.ctx.loglik_async = real
    assert T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0, comm=comm) == b
    ts.close()
    # threshold-fusion objective through the communicator: chunk-aligned shards, same value as the single-GPU call
    ts = comm.shard_trackset(lst, chunk=2000)
    a_th = T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0, comm=comm, fusion="threshold")
    b_th = T.cum_Proba_Cs(p, ts, 0.02, [1], None, 2, 1, 6, verbose=0, fusion="threshold")
    ts.close()
    assert a_th == b_th and a_th != a, (a_th, b_th, a)
    pa = T.predict_Bs(tr, 0.02, p, cell_dims=[1], nb_states=2, frame_len=5, comm=comm)
    pb = T.predict_Bs(tr, 0.02, p, cell_dims=[1], nb_states=2, frame_len=5)
    assert all(np.array_equal(pa[k], pb[k]) for k in pb)
    # zero-copy attach of torch tensors (device pointers) gives the same objective as the host-upload path
    from extrack_amd import _lib
    ctx = _lib.Context(0)
    tens = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in lst]
    for t_ in tens:
        ctx.attach_bucket(t_)
    torch.cuda.synchronize()
    ts2 = T.TrackSet(lst)
    model = T._objective_model(p, ts2, 0.02, [1], None, 2, 1, 6, 1)
    assert ctx.loglik(model) == ts2.loglik(model)
    ts2.close(); ctx.close()
    print("RCCL_PATH_OK", a)
finally:
    dist.destroy_process_group()
