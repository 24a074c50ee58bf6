"""Multi-GPU data parallelism for the likelihood objective: one process per GPU (torch.distributed).

The reference's only parallelism is ``multiprocessing.Pool.map`` over track chunks with the per-chunk
results concatenated and summed (extrack/tracking.py:1061-1069).  Here every rank keeps a contiguous row
range of every length bucket resident on its own MI355X and an evaluation ends with ONE all-reduce of the
scalar log-likelihood (8 bytes, RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests).
``predict_Bs`` needs no collective.  The dataset-global ``min_len`` / ``max_len`` (which decide the
``isBL`` flag of each bucket and the start of the stay-in-FOV term, tracking.py:1009-1010) are agreed on
with MIN/MAX all-reduces at shard time, never recomputed from the local shard.
"""
import numpy as np


def shard_range(n, rank, world, chunk=None):
    """Contiguous rows [start, stop) of a bucket of n tracks owned by `rank` (balanced to +-1).

    With ``chunk`` the unit of distribution is a whole chunk of ``chunk`` consecutive tracks: the threshold-fusion kernel
    takes its merge decisions per chunk (extrack/tracking.py:678-679, 1043), so a shard boundary inside a chunk would change
    the result; with chunk-aligned shards every rank evaluates exactly the chunks a single GPU would."""
    if chunk:
        nch = -(-int(n) // int(chunk))
        a, z = shard_range(nch, rank, world)
        return min(int(n), a * int(chunk)), min(int(n), z * int(chunk))
    base, rem = divmod(int(n), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class Comm:
    """Thin wrapper over an initialised torch.distributed process group."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = device if device is not None else ("cuda:%d" % torch.cuda.current_device() if self.backend == "nccl" else "cpu")
        self._buf = None

    # ---- scalars ------------------------------------------------------------------------------------------
    def allreduce_scalar(self, x, op="sum"):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return float(t.item())

    def global_min_max_len(self, lengths):
        lo = min(lengths) if len(lengths) else 1 << 30
        hi = max(lengths) if len(lengths) else 0
        return int(self.allreduce_scalar(lo, "min")), int(self.allreduce_scalar(hi, "max"))

    # ---- sharding -----------------------------------------------------------------------------------------
    def shard_buckets(self, tracks, sigmas=None, chunk=None):
        """Row-shards every bucket (in whole chunks when ``chunk`` is given); buckets whose local share is empty are dropped
        locally."""
        t_out, s_out = [], ([] if sigmas is not None else None)
        for i, b in enumerate(tracks):
            a, z = shard_range(len(b), self.rank, self.world, chunk)
            if z > a:
                t_out.append(b[a:z])
                if sigmas is not None:
                    s_out.append(sigmas[i][a:z])
        return t_out, s_out

    def shard_trackset(self, tracks, sigmas=None, device=0, chunk=None):
        from .engine import TrackSet
        lo, hi = self.global_min_max_len([b.shape[1] for b in tracks if len(b)])
        t_loc, s_loc = self.shard_buckets(tracks, sigmas, chunk)
        if not t_loc:
            raise ValueError("rank %d received no tracks: fewer tracks%s than ranks" % (self.rank, " (chunks)" if chunk else ""))
        ts = TrackSet(t_loc, s_loc, device=device, min_len=lo, max_len=hi)
        ts.shard_chunk = chunk  # chunk alignment of the shard boundaries (None: row-balanced)
        return ts

    # ---- posteriors: no collective in the data path, only an ordered gather of the per-rank row blocks ------------
    def gather_rows(self, local, dst=0):
        """``local``: {key: ndarray[rows_of_this_rank, ...]} with the same keys on every rank (rows follow ``shard_range``).
        Returns the row-concatenated dict on rank ``dst`` (rank order == original row order), None elsewhere."""
        parts = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(local, parts, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in local}

    def allreduce_loglik_th(self, ts, model, threshold, max_nb_states, chunk):
        """Threshold-fusion objective over chunk-aligned shards: local plan + apply kernels, all-reduce(sum) of the scalar."""
        if getattr(ts, "shard_chunk", None) != chunk:
            raise ValueError("fusion='threshold' needs shards aligned to the %d-track chunks (Comm.shard_trackset(..., chunk=%d))"
                             % (chunk, chunk))
        return self.allreduce_scalar(ts.loglik_th(model, threshold, max_nb_states, chunk), "sum")

    # ---- the per-evaluation collective ----------------------------------------------------------------------
    def allreduce_loglik(self, ts, model):
        """Local sum of log-likelihoods on this rank's GPU -> all-reduce(sum) -> python float."""
        if self.backend == "nccl":
            torch = self.torch
            if self._buf is None:
                self._buf = torch.zeros(1, dtype=torch.float64, device=self.device)
            stream = torch.cuda.current_stream()
            ts.ctx.set_stream(stream.cuda_stream)  # kernels and the collective are ordered on one stream
            ts.ctx.loglik_async(model, self._buf.data_ptr())
            self.dist.all_reduce(self._buf, op=self.dist.ReduceOp.SUM, group=self.group)
            return float(self._buf.item())
        return self.allreduce_scalar(ts.loglik(model), "sum")
