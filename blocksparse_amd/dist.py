"""Data-parallel use of the hot path: replicate layout tables and W, shard the minibatch N over ranks
(one process per GPU), sum the partial weight gradients with ONE all-reduce (RCCL over xGMI when the
backend is "nccl"; gloo on CPU for tests).  fprop/bprop need no communication: every minibatch column
is independent.  Replaces the reference's AllreduceNccl op for this path
(/root/reference/src/nccl_op.cc:166-201, blocksparse/nccl.py:27-56).
"""
import torch
import torch.distributed as dist


def shard_bounds(N, rank, world):
    """Contiguous, balanced split of N minibatch columns: the first N % world ranks get one extra."""
    q, r = divmod(N, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_minibatch(t, feature_axis, rank, world):
    """This rank's slice of an activation tensor: columns for axis 0 (C, N), rows for axis 1 (N, C)."""
    if feature_axis == 0:
        flat = t.reshape(t.shape[0], -1)
        lo, hi = shard_bounds(flat.shape[1], rank, world)
        return flat[:, lo:hi].contiguous()
    flat = t.reshape(-1, t.shape[-1])
    lo, hi = shard_bounds(flat.shape[0], rank, world)
    return flat[lo:hi].contiguous()


class DwAllReduce(object):
    """Sum partial dw over ranks, overlapped with whatever the caller enqueues next (normally bprop).

    start(dw) issues the collective asynchronously (for the NCCL/RCCL backend torch runs it on its own
    communication stream, ordered after the producing kernel through an event -- the same
    record-on-compute / wait-on-comm pattern as src/nccl_op.cc:513,168); wait() makes the current stream
    wait for it.  ``accumulate_fp32`` all-reduces an fp32 copy so the cross-rank sum is not rounded to
    16 bit per hop (RCCL supports bf16 too; the reference's op only took fp16/fp32, src/nccl_op.cc:140)."""

    def __init__(self, group=None, accumulate_fp32=False):
        self.group = group
        self.accumulate_fp32 = accumulate_fp32
        self._work = None
        self._buf = None
        self._dst = None

    def start(self, dw):
        if not dist.is_initialized():
            self._work = None
            return dw
        if self.accumulate_fp32 and dw.dtype != torch.float32:
            self._buf = dw.float()
            self._dst = dw
        else:
            self._buf = dw
            self._dst = None
        self._work = dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dw

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._dst is not None:
                self._dst.copy_(self._buf)
        self._buf = None
        self._dst = None
