"""A training step replayed from a hipGraph INSIDE the epocher loop (``semi_seg/epochers/new_pretrain.py:52-89`` is the
loop being mirrored): the ~90 kernel launches of a pre-train step cost ~4 ms of host time when issued one by one and
1.2 ms of GPU time, so the product loop -- not only the benchmark -- captures the step once and replays it.

What changes from step to step is HOST data: the label vectors the hooks build from the batch's strings, the per-sample
flip decisions drawn from the step's seed, and the loader's fresh image tensors.  They reach the replayed kernels
through persistent device memory:

* ``StepStage``: ONE persistent device block of slots.  A slot is bound to a *fill function* ``fill(batch) -> values`` by
  whoever consumes it (a hook binds its label vector, the epocher its flip flags); before every step all slots are
  refilled from the new batch on the host and uploaded by ``spcl_stage_bytes`` (the bytes travel as kernel arguments:
  one tiny launch, nothing to keep alive).  The captured kernels read the slots by pointer.
* the images are copied (view 1) / flipped (view 2) into a persistent ``[2n, C, H, W]`` buffer by ONE eager launch
  (``spcl_flip_pair``) in front of the replay: the loader's tensors may live anywhere.

``StepGraph`` is the small state machine around it: the first steps of a shape run eagerly (they are real steps, and
they warm the allocator), the next one is captured and replayed, every later one is refill + upload + replay.  With
``torch.distributed`` initialised the collective stays outside: compute graph, eager all-reduce, update graph -- and when
the compute phase calls ``StepGraph.cut(fn)`` on its way (ddp.FlatParams.reduce_early from the backward hook at the Conv3 |
Conv2 boundary) the compute graph ENDS there, ``fn`` (the early bucket's asynchronous all-reduce) runs eagerly, and a
second compute graph takes the rest of backward: three graphs, the early collective under blocks 2 .. 1.  Host-side
effects of a step that a replay would lose (meter adds of python floats) are logged at capture and re-applied.
Values that are baked into the capture (the age parameter gamma, changed once per epoch by
``SelfPacedINFONCEHook.__call__``) are part of the graph's key: a new epocher captures anew.  The capture also settles
the garbage collector (``_gc_settle``): an ~80 ms generation-2 collection in the middle of 1.2 ms steps is the one host
event that empties the GPU's queue."""
from __future__ import annotations

import ctypes
import os
import warnings

import numpy as np
import torch

from . import native as _n

_DTYPES = {"f32": np.float32, "u8": np.uint8, "i32": np.int32}


class StepStage:
    """persistent device block + host mirror of a step's host-written inputs (see the module docstring)"""

    def __init__(self, device, capacity: int = 64 * 1024):
        self.device = torch.device(device)
        self.capacity = capacity
        self._dev = torch.zeros(capacity, dtype=torch.uint8, device=self.device)
        self._host = np.zeros(capacity, dtype=np.uint8)
        self._slots = {}   # key -> (offset, count, numpy dtype, fill, device view)
        self._used = 0
        self.active = False  # True while a staged step runs: consumers bind / read their slots
        self.batch = None    # the current step's host-side description (what the fill functions receive)

    # ---- consumers
    def bind(self, key, count: int, kind: str, fill):
        """device view (``count`` elements of ``kind`` in f32 / u8 / i32) of slot ``key``, created on first use and filled
        from the current batch; ``fill(batch)`` returns ``count`` values.  Later steps refill it in ``begin``."""
        dt = _DTYPES[kind]
        slot = self._slots.get(key)
        if slot is not None and (slot[1] != count or slot[2] is not dt):
            raise RuntimeError(f"StepStage: slot {key!r} was bound with {slot[1]} x {slot[2].__name__}, now {count} x "
                               f"{dt.__name__} (a different batch shape needs its own stage)")
        if slot is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"StepStage: slot {key!r} first bound during graph capture (the eager steps before the "
                                   "capture must take the same path)")
            nbytes = (count * np.dtype(dt).itemsize + 15) // 16 * 16
            if self._used + nbytes > self.capacity:
                raise RuntimeError("StepStage: capacity exceeded")
            off = self._used
            self._used += nbytes
            tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dt]
            view = self._dev[off:off + count * np.dtype(dt).itemsize].view(tdt)
            slot = self._slots[key] = (off, count, dt, fill, view)
            self._write(slot)
            self._upload(off, nbytes)  # first use: this slot alone, in stream order ahead of its consumer
        return slot[4]

    # ---- the epocher
    def begin(self, batch, upload: bool = True):
        """a staged step starts: refill every slot from ``batch`` and upload the block (one launch per 3.5 KB).
        ``upload=False``: the caller uploads ``host_block()`` itself (spcl_flip_pair_stage carries it in its own launch)."""
        self.batch = batch
        self.active = True
        if self._slots:
            for slot in self._slots.values():
                self._write(slot)
            if upload:
                self._upload(0, self._used)

    def host_block(self):
        """(device pointer, host bytes, length) of the used part of the block -- for a launch that carries it along"""
        return self._dev.data_ptr(), self._host[:self._used], self._used

    def offset_of(self, key):
        slot = self._slots.get(key)
        return None if slot is None else slot[0]

    def end(self):
        self.active = False

    def _write(self, slot):
        off, count, dt, fill, _ = slot
        vals = np.asarray(fill(self.batch), dtype=dt).reshape(-1)
        if vals.size != count:
            raise RuntimeError(f"StepStage: fill returned {vals.size} values for a slot of {count}")
        self._host[off:off + count * vals.itemsize] = vals.view(np.uint8)

    def _upload(self, off, nbytes):
        src = self._host[off:off + nbytes]
        _n.call("spcl_stage_bytes", ctypes.c_void_p(self._dev.data_ptr() + off), src.ctypes.data_as(ctypes.c_void_p),
                nbytes, _n.stream())


_SIDE_STREAMS = {}
_CAPTURE_FAILED = set()  # device indices on which a step capture failed: later StepGraphs (one per epoch) do not retry
_WE_FROZE = False        # this module called gc.freeze(): only then may it gc.unfreeze() (the host application may freeze too)


_SETTLED_AT = 0.0        # time.monotonic() of the last FULL settle
_RESETTLE_AFTER_S = 120.0


def _gc_settle():
    """A replayed step costs the host ~150 us; ONE generation-2 garbage collection of a process with torch, the model and
    the data set loaded takes ~80 ms (measured: it lands inside a 30-step timed region every few runs and triples its mean;
    the GPU queue holds ~20 ms of work and runs dry).  At capture time collect, then move everything alive to the
    collector's permanent generation (``gc.freeze()``): later collections only look at what the steps themselves allocate.

    A capture happens once per EPOCH (the trainers build a new epocher every epoch) and the reference's epoch is 200 steps:
    the full settle -- undo the previous freeze so that dead cycles of earlier epochs are found, collect the whole heap,
    freeze again: 60 - 80 ms -- was a quarter of a 285 ms pre-train epoch (tools/diag/pretrain_epoch_time.py).  It now runs
    at the first capture and then at most every two minutes; the captures in between collect what is NOT frozen (the
    epochers, hooks and graphs of the epochs since: a few ms) and leave the permanent generation alone.  What died inside
    it is found by the next full settle.  ``SPCL_GC_FREEZE=0`` leaves the collector alone; ``gc_release()`` undoes it."""
    if os.environ.get("SPCL_GC_FREEZE", "1") == "0":
        return
    import gc
    import time
    global _WE_FROZE, _SETTLED_AT
    now = time.monotonic()
    if _WE_FROZE and now - _SETTLED_AT < _RESETTLE_AFTER_S:
        gc.collect()  # (the permanent generation is not traversed)
        return
    if _WE_FROZE:  # (never undo a freeze the host application made itself: gc.unfreeze() is process-wide)
        gc.unfreeze()
    gc.collect()
    gc.freeze()
    _WE_FROZE = True
    _SETTLED_AT = now


def gc_release(final: bool = True):
    """undo ``_gc_settle`` -- only a freeze this module made.  ``final=False`` is what an epocher says when its loop ends:
    the freeze STAYS for the next epoch's capture (``_gc_settle`` re-settles when it is due); the trainers say
    ``gc_release()`` when training ends, and so may anybody who drives epochers by hand."""
    global _WE_FROZE
    if os.environ.get("SPCL_GC_FREEZE", "1") == "0" or not _WE_FROZE or not final:
        return
    import gc
    gc.unfreeze()
    _WE_FROZE = False


_STEP_POOLS = {}


def step_pool(device=None):
    """the ONE graph memory pool per device that every step graph is captured into (``StepGraph._capture``).  A pool lives as
    long as a graph holds it (the allocator asserts on a handle whose last graph is gone): a one-launch anchor graph, never
    replayed, is captured into it once and kept."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    if dev not in _STEP_POOLS:
        handle = torch.cuda.graph_pool_handle()
        anchor = torch.cuda.CUDAGraph()
        side, cur = side_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            anchor.capture_begin(pool=handle)
            keep = torch.zeros(16, device=torch.device("cuda", dev))
            anchor.capture_end()
        cur.wait_stream(side)
        _STEP_POOLS[dev] = (handle, anchor, keep)
    return _STEP_POOLS[dev][0]


def side_stream(device=None):
    """the ONE private stream per device on which everything a step graph may later capture runs: warm steps, captures,
    and the eager steps of an epocher that graphs its steps (a ragged batch, a hook without a key).  autograd's
    AccumulateGrad nodes remember the stream they were created on; a node born on the default stream makes a later
    capture fork into it (hipStreamEndCapture segfaults on this stack)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    if dev not in _SIDE_STREAMS:
        _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[dev]


def run_on_side_stream(fn, device=None):
    """``fn()`` on ``side_stream`` between two stream waits (the caller's stream sees its results)"""
    side = side_stream(device)
    cur = torch.cuda.current_stream()
    if cur == side:
        return fn()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = fn()
    cur.wait_stream(side)
    return out


def collective_in_graph() -> bool:
    """SPCL_GRAPH_COLLECTIVE=1: a distributed job captures its collective INSIDE the step's one hipGraph (RCCL's kernels are
    capturable) instead of keeping it between a compute and an update graph.  Off by default: measured on ONE GPU only (a
    one-rank RCCL group, tools/diag/split_graph_cost.py) -- no multi-GPU node was available to check a captured ring."""
    return os.environ.get("SPCL_GRAPH_COLLECTIVE", "0") == "1"


def graph_default() -> bool:
    """the epochers capture their step unless SPCL_STEP_GRAPH=0"""
    return os.environ.get("SPCL_STEP_GRAPH", "1") != "0"


class StepGraph:
    """Capture-and-replay state machine of one epocher's step.

    ``compute()`` = forward + loss + backward + gradient gather, ``exchange()`` = the collective (eager, never
    captured), ``update()`` = optimizer + meters; ``split`` keeps ``exchange`` between two graphs (N > 1), otherwise the
    whole step is one graph.  ``run(key)`` executes one step: eagerly while ``warm`` steps of this ``key`` (batch shape +
    everything baked into the kernels' arguments) have not run yet, then capture + replay, then replay."""

    def __init__(self, compute, exchange, update, split: bool, warm: int = 2):
        self._compute, self._exchange, self._update = compute, exchange, update
        self._split, self._warm = bool(split), int(warm)
        self.key = None
        self._seen = 0
        self._graphs = None
        self._result = None
        self._host_log = []
        self.failed = False
        self.replays = 0
        # warm steps AND the capture run on one private stream: autograd's AccumulateGrad nodes remember the stream they
        # were created on, and a node kept alive from an eager step on the default stream would make the capture wait on
        # (i.e. fork into) the default stream
        self._stream = None
        self._cuts = []         # eager callables between the compute graphs, in order (recorded by ``cut`` at capture)
        self._capturing = None  # the compute graphs of the capture in progress

    def cut(self, fn):
        """Called from INSIDE ``compute()``: everything launched so far belongs in front of ``fn()``, everything after it
        behind.  In an eager step (or a one-process job, whose whole step is one graph) that is just ``fn()``.  While a
        split capture is in progress the current graph ends here and a new graph of the same memory pool begins; ``fn`` is
        recorded and runs at this point of every replay (the first of which follows the capture at once) -- NOT during the
        capture itself: a capture executes no kernel, what ``fn`` would communicate does not exist yet.  The caller may be
        autograd's device thread (a backward hook): the captures of a split step are therefore begun in hip's *relaxed*
        mode, the only one in which a capture may be ended by another thread than the one that began it (measured on this
        stack, tools/diag/graph_split_in_backward.py: `thread_local` and `global` fail with
        hipErrorStreamCaptureWrongThread)."""
        if self._capturing is None:
            return fn()
        self._capturing[-1].capture_end()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=step_pool(), capture_error_mode="relaxed")
        self._capturing.append(g)
        self._cuts.append(fn)
        return None

    @property
    def captured(self):
        return self._graphs is not None

    def _eager(self):
        loss = self._compute()
        self._exchange()
        self._update(loss)
        return loss

    def _warm_step(self):
        if self._stream is None:  # ONE private stream per device for every StepGraph of the process (epoch after epoch)
            self._stream = side_stream()
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            loss = self._eager()
        cur.wait_stream(self._stream)
        return loss

    def run(self, key):
        from .contrastyou import meters as _meters
        if not self.failed and torch.cuda.current_device() in _CAPTURE_FAILED:
            self.failed = True  # an earlier epocher's capture failed on this device: stay eager, do not capture again
        if self.failed:
            # eager, but on the SAME private stream as every other backward pass of this model: autograd's AccumulateGrad
            # nodes remember their stream, and a later capture (another model, another shape) must not fork into the
            # caller's stream (module docstring)
            return self._warm_step()
        if key != self.key:  # a new shape / configuration: drop the graphs, start over
            self.key, self._seen, self._graphs, self._result = key, 0, None, None
        if self._graphs is None:
            if self._seen < self._warm:
                self._seen += 1
                return self._warm_step()
            try:
                self._capture()
            except Exception as e:  # noqa: BLE001 -- a capture that fails must not end the training run
                self.failed = True
                _CAPTURE_FAILED.add(torch.cuda.current_device())
                self._graphs = None
                # the python floats the hooks handed to the meters during the failed capture were applied once already and
                # the eager step below hands them over again: take the first application back
                for meter, value, n in self._host_log:
                    meter.retract(value, n)
                self._host_log = []
                try:
                    torch.cuda.synchronize()
                except Exception:  # noqa: BLE001
                    pass
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                              "continuing with eager launches (this process will not try to capture on this device again)")
                return self._warm_step()
            self._replay(first=True)
            return self._result
        self._replay(first=False)
        return self._result

    def _capture(self):
        from .contrastyou import meters as _meters
        self._host_log = []
        _gc_settle()
        torch.cuda.synchronize()
        state = {}
        _meters.begin_host_log()
        # Captures are begun and ended by hand, not by ``torch.cuda.graph``: (1) ``cut`` may end a graph half way through
        # ``compute()``; (2) the context manager collects garbage (done above, cheaply) and EMPTIES THE ALLOCATOR'S CACHE --
        # with a new epocher, i.e. a new capture, every 200 steps that was 6 - 7 ms per epoch of hipFree / hipMalloc for the
        # same ~1.5 GB of activations (tools/diag/capture_cost.py: graph.__enter__ 9.5 ms of a 14 ms capture step).  All
        # step graphs of a device share ONE memory pool instead: the blocks the previous epoch's graph gave back when it was
        # collected are the blocks this capture takes.
        mode = "relaxed" if self._split else "global"
        pool = step_pool()

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode=mode)
            return g

        def capture(body, graphs):
            """``body()`` captured into ``graphs`` (a list that ``cut`` may extend through ``self._capturing``)"""
            try:
                body()
            except BaseException:
                try:  # leave the stream out of capture mode whatever happened
                    graphs[-1].capture_end()
                except Exception:  # noqa: BLE001
                    pass
                raise
            graphs[-1].capture_end()

        try:
            with torch.cuda.stream(self._stream):
                if self._split:
                    self._cuts = []
                    self._capturing = [begin()]
                    try:
                        capture(lambda: state.__setitem__("loss", self._compute()), self._capturing)
                    finally:
                        compute_graphs, self._capturing = self._capturing, None
                    gb = [begin()]
                    capture(lambda: self._update(state["loss"]), gb)
                    self._graphs = (*compute_graphs, gb[0])
                else:
                    g = [begin()]

                    def whole():
                        state["loss"] = self._compute()
                        self._exchange()  # (a no-op in a one-process job; see ``collective_in_graph``)
                        self._update(state["loss"])
                    capture(whole, g)
                    self._graphs = (g[0],)
        finally:
            self._host_log = _meters.end_host_log()
        self._result = state["loss"]

    def _replay(self, first):
        if self._split:
            for i, g in enumerate(self._graphs[:-1]):
                g.replay()
                if i < len(self._cuts):
                    self._cuts[i]()
            self._exchange()
            self._graphs[-1].replay()
        else:
            self._graphs[0].replay()
        if not first:  # the capture pass already performed the host-side adds once
            for meter, value, n in self._host_log:
                meter.add(value, n)
        self.replays += 1
