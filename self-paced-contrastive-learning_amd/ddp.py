"""Data-parallel exchange step of the pre-train path: ONE flat gradient bucket all-reduced per step.

The reference is single-process; the path shards by batch, so the only collective is the gradient sum:
1 311 056 fp32 values = 5.24 MB per pre-train step, one RCCL all-reduce over xGMI (latency-bound; a single bucket
avoids per-tensor launches).  BatchNorm statistics and contrastive negatives stay per-GPU.  Backend-agnostic
(``nccl`` == RCCL on ROCm; ``gloo`` in the CPU tests)."""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def on_master() -> bool:
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0


class GradBucket:
    """Flat fp32 bucket over the gradients of ``params`` (fixed order).  ``allreduce()`` averages them across ranks
    and leaves ``p.grad`` pointing at views of the bucket (no copy back)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert len(self.params) > 0
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.group = process_group

    @property
    def nbytes(self):
        return self.numel * 4

    def arm_sinks(self):
        """Offer every parameter its slice of the bucket as the place its next gradient is written to
        (functional.take_grad_sink, claimed in backward): the HIP backward kernels then fill the bucket directly and
        ``gather`` has nothing left to copy.  Call after the gradients were cleared (``p.grad = None``), once per step,
        before backward.  Also opens the deferred weight-gradient queue (functional.DeferredWgrads): the wide layers'
        weight gradients are written into their slices by one batched launch when ``gather`` flushes it."""
        from . import functional as _F
        for p, v in zip(self.params, self.views):
            p._grad_sink = v
            p._grad_sink_armed = True
        if self.flat.is_cuda:
            _F.open_deferred_wgrads()

    def disarm_sinks(self):
        for p in self.params:
            p._grad_sink_armed = False

    def gather(self):
        """grads -> bucket (one fused foreach copy of those not already written in place); params without a grad
        contribute zeros.  Flushes the deferred weight gradients first and disarms the sinks nobody claimed."""
        from . import functional as _F
        deferred = _F.flush_deferred_wgrads()
        self.disarm_sinks()
        srcs, dsts = [], []
        for p, v in zip(self.params, self.views):
            if v.data_ptr() in deferred:
                # the slice holds the batched launch's result; autograd holds what OTHER uses of the parameter gave
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.add_(p.grad)
                p.grad = v
            elif p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                srcs.append(p.grad)
                dsts.append(v)
        if dsts:
            torch._foreach_copy_(dsts, srcs)

    def allreduce(self):
        self.gather()
        if is_distributed():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat


class FlatParams(GradBucket):
    """All trainable parameters as views of ONE flat fp32 tensor, optimised as a single ``nn.Parameter``.

    The optimizer (torch RAdam, host code per the north star) then runs ~10 fused kernels per step instead of
    ~10 per parameter tensor, and the flat gradient bucket it consumes is exactly the buffer the RCCL all-reduce
    works on.  Module ``state_dict``s are unaffected (the parameters keep their identity, only their storage moves)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        super().__init__(params, process_group)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=self.flat.device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                v = self.data[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                off += p.numel()
        self.param = torch.nn.Parameter(self.data)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self.param.grad = None
        self.arm_sinks()

    def gather_grads(self):
        """module grads -> flat bucket, which becomes ``self.param.grad`` (no communication)."""
        self.gather()
        self.param.grad = self.flat
        return self.flat

    def allreduce_(self):
        """mean of the flat bucket across ranks, in place (the step's ONE collective; no-op for a single process)."""
        if is_distributed():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))
        return self.flat

    def reduce(self):
        """module grads -> flat bucket -> (all-reduce mean across ranks) -> ``self.param.grad``."""
        self.gather_grads()
        return self.allreduce_()


@torch.no_grad()
def broadcast_state(*modules: torch.nn.Module, src: int = 0):
    """Rank ``src``'s parameters and buffers (BN running statistics) to every rank, once, before training."""
    if not is_distributed():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src)
