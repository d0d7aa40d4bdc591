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

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, allow_missing_grads: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert len(self.params) > 0
        # A member that received NO gradient would still be stepped by an optimizer over the flat parameter (zeros in
        # its slice: weight decay and stale momentum move it), where torch.optim -- and the reference's optimiser over
        # model.parameters() -- skips it.  So the bucket must hold exactly the parameters the step reaches: build it
        # inside ``model.set_grad(False, start=until, include_start=False)`` (as the trainers do).  ``gather`` raises
        # otherwise, unless the caller asks for the zeros explicitly.
        self.allow_missing_grads = allow_missing_grads
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
        # a FRESH queue per arming: whatever an aborted step left queued (and the tensors it kept alive) is dropped
        self._queue = _F.DeferredWgrads() if self.flat.is_cuda else None
        for p, v in zip(self.params, self.views):
            p._grad_sink = v
            p._grad_sink_armed = True
            v._spcl_queue = self._queue

    def disarm_sinks(self):
        for p, v in zip(self.params, self.views):
            p._grad_sink_armed = False
            v._spcl_queue = None
        self._queue = None

    def gather(self, first: int = 0, last: int = None, final: bool = True):
        """grads -> bucket (one fused foreach copy of those not already written in place) for members
        ``first .. last - 1`` (default: all).  Flushes the deferred weight gradients first; ``final`` (the step's last
        gather) also closes their queue and disarms the sinks nobody claimed.  A member without a gradient raises (see
        the constructor) or, with ``allow_missing_grads``, contributes zeros."""
        queue = getattr(self, "_queue", None)
        deferred = set()
        if queue is not None:  # this bucket's own queue: its targets are slices of THIS bucket only
            queue.flush()
            deferred = set(queue.targets)
        if final:
            self.disarm_sinks()
            deferred = deferred | getattr(self, "_early_deferred", set())
            self._early_deferred = set()
        else:
            self._early_deferred = getattr(self, "_early_deferred", set()) | deferred
        last = len(self.params) if last is None else last
        srcs, dsts = [], []
        for p, v in zip(self.params[first:last], self.views[first:last]):
            if v.data_ptr() in deferred:
                # the slice holds the batched launch's result; autograd holds what OTHER uses of the parameter gave
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.add_(p.grad)
                p.grad = v
            elif p.grad is None:
                if not self.allow_missing_grads:
                    raise RuntimeError(
                        f"GradBucket.gather: member {tuple(p.shape)} received no gradient in this step; an optimizer "
                        "over the flat parameter would still move it (torch.optim would skip it).  Build the bucket "
                        "from the parameters the step reaches, or pass allow_missing_grads=True to get zeros.")
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

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, allow_missing_grads: bool = False):
        super().__init__(params, process_group, allow_missing_grads)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=self.flat.device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                v = self.data[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                off += p.numel()
        self.param = torch.nn.Parameter(self.data)
        self._early_idx, self._early_off, self._early = None, 0, None
        self.cutter = None  # see ``reduce_early``
        # fold_mean: ``allreduce_`` leaves the ranks' SUM in the bucket and the factor 1 / world in ``grad_scale`` for an
        # optimizer that applies it while it streams the bucket anyway (optim.FusedRAdam.step(grad_scale=)): one launch
        # and one pass over the bucket less per step.  Off: the bucket holds the mean (``div_``), grad_scale stays 1.
        self.fold_mean = False

    @property
    def grad_scale(self) -> float:
        """the factor a consumer of the bucket must apply to obtain the ranks' MEAN: 1 / world while ``fold_mean`` is on
        in a distributed job, else exactly 1.0.  A pure function of (fold_mean, world size): valid BEFORE the first
        collective, so a captured update graph that bakes it in (stepgraph.py never runs the exchange inside the capture)
        holds the right value whatever ran before the capture."""
        if self.fold_mean and is_distributed():
            return 1.0 / dist.get_world_size(self.group)
        return 1.0

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self.param.grad = None
        self._early = None  # a step that never reached allreduce_() must not leave the early bucket marked as sent
        self.arm_sinks()

    def gather_grads(self):
        """module grads -> flat bucket, which becomes ``self.param.grad`` (no communication).  With the early bucket
        already on its way (``reduce_early``) only the remaining head is gathered."""
        if self._early is not None:
            self.gather(first=0, last=self._early_idx, final=True)
        else:
            self.gather()
        self.param.grad = self.flat
        return self.flat

    def allreduce_(self):
        """mean of the flat bucket across ranks, in place (the step's ONE collective, or -- with the early bucket on its
        way -- the head's plus the wait for the tail's; no-op for a single process)."""
        if is_distributed():
            # On the CALLER's stream (round 6; until then on a private communication stream between two stream waits).  A
            # blocking collective of this torch is enqueued on the current stream itself; the private stream put two event
            # hand-overs between the step's hipGraph replays, and a graph replay that starts behind another stream's event
            # starts late: one-rank RCCL group on an MI355X, whole step, 0.989 ms as one graph -> 1.071 ms as compute graph |
            # collective on the private stream | update graph -> 1.012 ms with the collective where the graphs are
            # (tools/diag/split_graph_cost.py, profiles/r06_experiments/NOTES.md).
            if self._early is not None:
                dist.all_reduce(self.flat[:self._early_off], op=dist.ReduceOp.SUM, group=self.group)
                if self._early is not True:
                    self._early.wait()
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if not self.fold_mean:  # (fold_mean: the bucket keeps the SUM, ``grad_scale`` is 1 / world)
                self.flat.div_(dist.get_world_size(self.group))
        self._early = None
        return self.flat

    def reduce(self):
        """module grads -> flat bucket -> (all-reduce mean across ranks) -> ``self.param.grad``."""
        self.gather_grads()
        return self.allreduce_()

    # ---- two-bucket overlap (off unless ``overlap_from`` is called).  Members are in module order, so the
    # gradients that backward finishes FIRST (projector, Conv5 .. Conv3) are the TAIL of the flat bucket:
    # ``reduce_early()`` -- called from a backward hook at the Conv3 | Conv2 boundary -- gathers that tail and starts its
    # all-reduce asynchronously while Conv2 .. Conv1 are still being differentiated; ``reduce()`` then handles the head
    # and waits.  Sums are elementwise, so the result equals the one-bucket reduce (tests/test_ddp_gloo.py).
    def overlap_from(self, first_early_param):
        """members from ``first_early_param`` (a member, or its index) to the end form the early bucket"""
        idx = first_early_param if isinstance(first_early_param, int) else \
            next(i for i, p in enumerate(self.params) if p is first_early_param)
        assert 0 < idx < len(self.params)
        self._early_idx = idx
        self._early_off = sum(p.numel() for p in self.params[:idx])
        return self

    def reduce_early(self):
        """gather the early bucket and start its all-reduce (no-op when overlap is off or it already ran this step).  With
        a ``cutter`` (stepgraph.StepGraph.cut of the step graph that replays this bucket's step) the collective's start is
        handed to it: the gather is the last thing of one hipGraph, the start runs eagerly behind that graph's replay, the
        rest of backward is the next graph."""
        if self._early_idx is None or self._early is not None:
            return
        self.gather(first=self._early_idx, final=False)
        self._early = True  # (host state: ``gather_grads`` takes the head only -- also in a capture, where nothing is sent)
        if self.cutter is not None:
            self.cutter(self._start_early)
        else:
            self._start_early()

    def _start_early(self):
        self._early = True
        if is_distributed():
            self._early = dist.all_reduce(self.flat[self._early_off:], op=dist.ReduceOp.SUM, group=self.group,
                                          async_op=True)

    def early_hook(self):
        """a callback for a backward hook (``tensor.register_hook``) that starts the early bucket"""
        def hook(*args):
            self.reduce_early()
        return hook


def enable_unet_overlap(flat: "FlatParams", net, early_block: str = "Conv3", hook_block: str = "Conv2"):
    """Two-bucket overlap for a UNet encoder step: everything from ``early_block`` on (Conv3 .. Conv5 and
    whatever follows the encoder in the bucket: projector heads) is all-reduced from a backward hook at the output of
    ``hook_block`` -- it fires when the gradient of that block's pooled output exists, i.e. when every later block has
    been differentiated.  ``disable_unet_overlap`` undoes it."""
    first = next(getattr(net, "_" + early_block).parameters())
    flat.overlap_from(first)
    net._boundary_hooks[hook_block] = flat.early_hook()


def disable_unet_overlap(flat: "FlatParams", net):
    net._boundary_hooks.clear()
    flat._early_idx, flat._early_off, flat._early = None, 0, None


@torch.no_grad()
def broadcast_state(*modules: torch.nn.Module, src: int = 0):
    """Rank ``src``'s parameters and buffers (BN running statistics) to every rank, once, before training."""
    if not is_distributed():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src)
