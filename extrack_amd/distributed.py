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
    """Contiguous rows [start, stop) of ONE bucket of n tracks owned by `rank` (balanced to +-1 row, or +-1 chunk with ``chunk``).
    Used where every bucket is split on its own (``predict_Bs``: rank order == row order); the objective's shards come from
    ``shard_plan``, which balances the remainders over all buckets."""
    if chunk:
        nch = -(-int(n) // int(chunk))
        a, z = shard_range(nch, rank, world)
        return min(int(n), a * int(chunk)), min(int(n), z * int(chunk))
    base, rem = divmod(int(n), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_plan(sizes, lengths, world, chunk=None):
    """Row ranges of every bucket for every rank: ``plan[b][r] = (start, stop)``, contiguous and in rank order inside a bucket.

    The unit of distribution is one track, or - with ``chunk`` - a whole chunk of ``chunk`` consecutive tracks: the
    threshold-fusion kernel takes its merge decisions per chunk (extrack/tracking.py:678-679, 1043), so a shard boundary inside a
    chunk would change the result; with chunk-aligned shards every rank evaluates exactly the chunks a single GPU would.
    Every bucket gives ``units // world`` units to each rank; the remaining ``units % world`` go, one each, to the ranks with
    the least accumulated work (tracks x steps) so far.  A dataset of many small buckets (one chunk each) is thereby dealt
    round-robin instead of piling up on rank 0, and equal buckets end up balanced to one unit overall, not one unit per bucket.
    Deterministic: every rank computes the same plan from the same (sizes, lengths)."""
    world = int(world)
    load = [0.0] * world
    plan = []
    for n, L in zip(sizes, lengths):
        n = int(n)
        unit = int(chunk) if chunk else 1
        units = -(-n // unit)
        base, rem = divmod(units, world)
        extra = set(sorted(range(world), key=lambda r: (load[r], r))[:rem])
        ranges, pos = [], 0
        for r in range(world):
            cnt = base + (1 if r in extra else 0)
            a, z = min(n, pos * unit), min(n, (pos + cnt) * unit)
            ranges.append((a, z))
            load[r] += (z - a) * max(int(L) - 1, 1)
            pos += cnt
        plan.append(ranges)
    return plan


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
        self._pin = {}

    # ---- scalars ------------------------------------------------------------------------------------------
    def allreduce_scalar(self, x, op="sum"):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return float(t.item())

    def allreduce_vector(self, v, op="sum"):
        """Reduction of a small fp64 vector over the ranks (objective + gradient: 1 + nvar doubles per evaluation)."""
        t = self.torch.as_tensor(np.asarray(v, dtype=np.float64), device=self.device).clone()
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t.cpu().numpy()

    def allreduce_checked(self, compute, n):
        """Sum over the ranks of the ``n`` doubles ``compute()`` returns locally, with error agreement: an exception on ANY rank (a
        capacity limit hit only by that rank's longest bucket, an out-of-memory, ...) is carried through the same collective as a
        flag and raised on EVERY rank afterwards - a rank raising alone would leave its peers blocked in the all-reduce."""
        err = None
        try:
            v = np.asarray(compute(), dtype=np.float64).reshape(n)
        except Exception as e:  # noqa: BLE001 - re-raised below, after the collective
            err, v = e, np.zeros(n)
        out = self.allreduce_vector(np.concatenate([v, [0.0 if err is None else 1.0]]))
        if out[-1] > 0:
            if err is not None:
                raise err
            raise RuntimeError("the evaluation failed on %d other rank(s)" % int(round(out[-1])))
        return out[:-1]

    def global_min_max_len(self, lengths):
        lo = min(lengths) if len(lengths) else 1 << 30
        hi = max(lengths) if len(lengths) else 0
        return int(self.allreduce_scalar(lo, "min")), int(self.allreduce_scalar(hi, "max"))

    # ---- sharding -----------------------------------------------------------------------------------------
    def shard_buckets(self, tracks, sigmas=None, chunk=None, dts=None):
        """Row-shards every bucket per ``shard_plan`` (whole chunks when ``chunk`` is given); buckets whose local share is empty
        are dropped locally (a rank may end up with no bucket at all: it then contributes 0 to the objective).  ``dts`` (per-track
        time steps [N_l, l]) are cut like the tracks; returns (tracks, sigmas) or, with ``dts``, (tracks, sigmas, dts)."""
        plan = shard_plan([len(b) for b in tracks], [b.shape[1] for b in tracks], self.world, chunk)
        t_out, s_out, d_out = [], ([] if sigmas is not None else None), ([] if dts is not None else None)
        for i, b in enumerate(tracks):
            a, z = plan[i][self.rank]
            if z > a:
                t_out.append(b[a:z])
                if sigmas is not None:
                    s_out.append(sigmas[i][a:z])
                if dts is not None:
                    d_out.append(dts[i][a:z])
        return (t_out, s_out) if dts is None else (t_out, s_out, d_out)

    def agree(self, ok, what="an operation"):
        """Collective error agreement: every rank passes its local success flag; if ANY rank failed, all ranks raise here, so that
        no rank walks on into a collective its peers will never join."""
        if self.allreduce_scalar(1.0 if ok else 0.0, "min") < 0.5:
            raise RuntimeError("%s failed on at least one rank%s" % (what, "" if ok else " (this one: rank %d)" % self.rank))

    def local_device(self):
        """GPU index this rank's kernels run on: the device of the communicator (nccl), else 0."""
        d = str(self.device)
        return int(d.split(":")[1]) if d.startswith("cuda:") else 0

    def shard_trackset(self, tracks, sigmas=None, device=None, chunk=None, dts=None):
        """Uploads this rank's shard of ``tracks`` (the WHOLE dataset, same list on every rank) to its GPU.  ``min_len`` /
        ``max_len`` are the dataset-global ones.  A rank left without tracks gets an empty TrackSet (objective 0.0).  ``dts``:
        per-track time steps [N_l, l] per bucket (threshold-fusion objective; needs ``chunk``: the field-of-view table of a chunk
        comes from the time steps of ITS tracks, tracking.py:507-511, so the shards must be whole chunks)."""
        from .engine import TrackSet
        lens = [b.shape[1] for b in tracks if len(b)]
        lo, hi = self.global_min_max_len(lens)
        if device is None:
            device = self.local_device()
        ts, err = None, None
        try:
            if not lens:
                raise ValueError("No track could be detected. The loaded tracks seem empty. Errors often come from wrong input paths.")
            if dts is not None and chunk is None:
                raise ValueError("per-track time steps need chunk-aligned shards: pass chunk=max_number_of_tracks_per_matrix")
            d_loc = None
            if dts is None:
                t_loc, s_loc = self.shard_buckets(tracks, sigmas, chunk)
            else:
                t_loc, s_loc, d_loc = self.shard_buckets(tracks, sigmas, chunk, dts)
            ts = TrackSet(t_loc, s_loc, device=device, min_len=lo, max_len=hi, allow_empty=True, dts=d_loc)
            ts.shard_chunk = chunk  # chunk alignment of the shard boundaries (None: row-balanced)
        except Exception as e:  # agree before raising: a lone raising rank would leave the others blocked in the next collective
            err = e
        try:
            self.agree(err is None, "sharding the dataset")
        except RuntimeError:
            if ts is not None:
                ts.close()
            raise err if err is not None else RuntimeError("sharding the dataset failed on another rank")
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
        if self.backend == "nccl":
            return float(self._reduce_on_stream(ts, lambda ptr: ts.ctx.loglik_th_async(model, threshold, max_nb_states, chunk, ptr))[0])
        return float(self.allreduce_checked(lambda: [ts.loglik_th(model, threshold, max_nb_states, chunk) if ts.n_tracks else 0.0], 1)[0])

    def _reduce_on_stream(self, ts, enqueue, n=1):
        """Stream-ordered reduction of ``n`` doubles: ``enqueue(device_ptr)`` launches this rank's kernels, which leave the local sums in
        the first ``n`` slots of a device buffer; slot ``n`` is the failure flag of ``allreduce_checked`` (set when the enqueue raised);
        RCCL reduces all of it on the same stream and one small read comes back - no host round trip before the collective."""
        buf = self._device_buffer(ts, n + 1)
        err = None
        try:
            if ts.n_tracks:
                enqueue(buf.data_ptr())
            else:
                buf[0:n].zero_()
        except Exception as e:  # noqa: BLE001 - re-raised on every rank after the collective
            err = e
            buf[0:n].zero_()
            buf[n:n + 1].fill_(1.0)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        h = self._read_back(buf, n + 1)
        if h[n] > 0:
            buf[n:n + 1].zero_()
            if err is not None:
                raise err
            raise RuntimeError("the evaluation failed on %d other rank(s)" % int(round(float(h[n]))))
        return h[:n].copy()

    def _read_back(self, buf, n):
        """The reduced doubles on the host: an asynchronous copy into a pinned buffer allocated once per size + ONE stream
        synchronisation (``buf.cpu()`` allocates a tensor and takes the slow pageable path every evaluation: at the strong-scaled
        shard of 125 000 tracks the kernel is ~0.4 ms and every 10 us of host work per evaluation is 2.5 % of the step)."""
        torch = self.torch
        if not buf.is_cuda:
            return buf.numpy().copy()
        pin = self._pin.get(n)
        if pin is None:
            t = torch.empty(n, dtype=torch.float64, pin_memory=True)
            pin = self._pin[n] = (t, t.numpy())
        pin[0].copy_(buf, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return pin[1]

    def allreduce_loglik_grad(self, ts, model, tangents, n_dir):
        """{sum LL, d sum LL / d theta} over the ranks: with RCCL the gradient kernels write their 1 + n_dir sums into the buffer the
        collective reduces, on one stream (extrack_loglik_grad_async); with gloo (CPU tests) through the host."""
        if self.backend == "nccl":
            return self._reduce_on_stream(ts, lambda ptr: ts.ctx.loglik_grad_async(model, tangents, ptr), 1 + n_dir)

        def local():
            if not ts.n_tracks:
                return np.zeros(1 + n_dir)
            ll, g = ts.ctx.loglik_grad(model, tangents)
            return np.concatenate([[ll], g])
        return self.allreduce_checked(local, 1 + n_dir)

    def allreduce_loglik_th_grad(self, ts, model, tangents, n_dir, threshold, max_nb_states, chunk):
        """{sum LL, d sum LL / d theta} of the threshold-fusion objective at the frozen plan of the evaluation, over chunk-aligned shards
        (every rank plans and differentiates exactly the chunks a single GPU would): one all-reduce of 1 + n_dir doubles."""
        if getattr(ts, "shard_chunk", None) != chunk:
            raise ValueError("fusion='threshold' needs shards aligned to the %d-track chunks (Comm.shard_trackset(..., chunk=%d))"
                             % (chunk, chunk))
        if self.backend == "nccl":
            return self._reduce_on_stream(ts, lambda ptr: ts.ctx.loglik_th_grad_async(model, tangents, threshold, max_nb_states, chunk, ptr), 1 + n_dir)

        def local():
            if not ts.n_tracks:
                return np.zeros(1 + n_dir)
            ll, g = ts.ctx.loglik_th_grad(model, tangents, threshold, max_nb_states, chunk)
            return np.concatenate([[ll], g])
        return self.allreduce_checked(local, 1 + n_dir)

    def _device_buffer(self, ts, n=2):
        """fp64 device buffer of ``n`` doubles on this rank's GPU (the last one is the failure flag) + the context bound to torch's
        current stream (kernels and the collective are ordered on one stream)."""
        torch = self.torch
        if self._buf is None:
            self._buf = {}
        buf = self._buf.get(n)
        if buf is None:
            buf = self._buf[n] = torch.zeros(n, dtype=torch.float64, device=self.device)
        if ts.n_tracks and buf.device.index != ts.ctx.device:
            raise RuntimeError("the TrackSet lives on GPU %d but the communicator reduces on %s: pass device=%d (LOCAL_RANK) when "
                               "building it" % (ts.ctx.device, buf.device, buf.device.index))
        if ts.n_tracks:
            ts.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        return buf

    # ---- the per-evaluation collective ----------------------------------------------------------------------
    def allreduce_loglik(self, ts, model):
        """Local sum of log-likelihoods on this rank's GPU -> all-reduce(sum) -> python float."""
        if self.backend == "nccl":
            return float(self._reduce_on_stream(ts, lambda ptr: ts.ctx.loglik_async(model, ptr))[0])
        return float(self.allreduce_checked(lambda: [ts.loglik(model) if ts.n_tracks else 0.0], 1)[0])
