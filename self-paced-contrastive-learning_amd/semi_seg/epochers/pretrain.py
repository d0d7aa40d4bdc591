"""Mirror of the pre-train epocher: ``EpocherBase`` (contrastyou/epochers/base.py:16-114), the hook-calling
``forward_pass``/``regularization`` wrappers (semi_seg/epochers/new_epocher.py:33-50) and the hot loop
``_PretrainEpocherMixin._run_pretrain/_forward_pass`` (semi_seg/epochers/new_pretrain.py:52-96).

Same control flow and hook wire format; what changes is WHERE work happens: no per-step ``.item()`` (meters take
device scalars), flips are batched gathers, the encoder/projector/loss are the HIP kernels, and with
``torch.distributed`` initialised the gradients go through one flat RCCL all-reduce (ddp.GradBucket) before the
optimizer step."""
import random
from typing import Iterable, Optional

import torch
from torch import nn

from ...contrastyou import meters as _meters
from ...contrastyou.meters import AverageValueMeter, MeterInterface
from ... import ddp as _ddp
from ... import native as _n
from ... import functional as F_hip
from ... import stepgraph as _sg
from .helper import FixRandomSeed, TensorRandomFlip


def unzip_twice_transformed(data, device):
    """``preprocess_input_with_twice_transformation`` (semi_seg/epochers/helper.py:27-36) + ``_unzip_data``
    (new_pretrain.py:98-102): ((image, image_tf, target, target_tf), filename, (partition, group))."""
    (image, image_tf, *_), filename, (partition_list, group_list) = data
    image = image.to(device, non_blocking=True)
    image_tf = image_tf.to(device, non_blocking=True)
    return (image, image_tf), None, filename, partition_list, group_list


def _cat_views(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``torch.cat([a, b])`` (new_pretrain.py:93) -- as a view when a and b already are the two halves of one buffer."""
    if (a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous() and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], dim=0)


class PretrainEncoderEpocher:
    meter_focus = "semi"

    def __init__(self, *, model: nn.Module, optimizer, chain_dataloader: Iterable, num_batches: int, cur_epoch=0,
                 device="cuda", inference_until: str = "Conv5", grad_bucket: Optional[_ddp.GradBucket] = None,
                 flat_params: Optional[_ddp.FlatParams] = None, graph: Optional[bool] = None, **kwargs) -> None:
        self._model = model
        self._optimizer = optimizer
        self._chain_dataloader = chain_dataloader
        self._num_batches = num_batches
        self._cur_epoch = cur_epoch
        self._device = torch.device(device)
        self._inference_until = inference_until
        self._affine_transformer = TensorRandomFlip(axis=[1, 2], threshold=0.8)
        self._grad_bucket = grad_bucket
        self._flat_params = flat_params  # optimizer steps ONE flat parameter (ddp.FlatParams) when given
        from ...optim import FusedRAdam
        if flat_params is not None:
            # FusedRAdam: the exchange leaves the ranks' SUM, 1 / world is applied inside the RAdam kernel; any other
            # optimizer reads the bucket as it is and must find the MEAN there (the flag is state of the shared FlatParams:
            # an earlier epocher may have set it)
            flat_params.fold_mean = isinstance(optimizer, FusedRAdam)
        self._hooks = []
        self.meters = MeterInterface(default_focus=self.meter_focus)
        with self.meters.focus_on(self.meter_focus):
            self.meters.register_meter("lr", AverageValueMeter())
            self.meters.register_meter("reg_loss", AverageValueMeter())
        self.cur_batch_num = 0
        self._ones = {}
        # the step as a hipGraph (stepgraph.py): on by default where it can work -- a CUDA device, the flat parameter with
        # the fused optimizer, hooks that declare what they bake into their launches (``graph_key``)
        self._graph_on = _sg.graph_default() if graph is None else bool(graph)
        self.stage = None       # stepgraph.StepStage: the step's host-written inputs (hooks bind their label slots)
        self._step_graph = None
        self._pair = None       # persistent [2n, C, H, W] input pair of the captured step
        self._staged = None

    # ---- contrastyou/epochers/base.py:47-60
    def add_hook(self, hook):
        self._hooks.append(hook)
        hook.set_epocher(self)

    def add_hooks(self, hooks):
        for h in hooks:
            self.add_hook(h)

    def close_hooks(self):
        for h in self._hooks:
            h.close()

    @staticmethod
    def on_master():
        return _ddp.on_master()

    def init(self):
        pass

    # ---- new_epocher.py:33-50
    def forward_pass(self, **kwargs):
        for h in self._hooks:
            h.before_forward_pass(**kwargs)
        result = self._forward_pass(**kwargs)
        for h in self._hooks:
            h.after_forward_pass(**kwargs, result_dict=result)
        return result

    def regularization(self, **kwargs):
        for h in self._hooks:
            h.before_regularization(**kwargs)
        result = self._regularization(**kwargs)
        for h in self._hooks:
            h.after_regularization(**kwargs, result_dict=result)
        return result

    def _regularization(self, **kwargs):  # new_epocher.py:234-238
        if len(self._hooks) > 0:
            losses = [h(**kwargs) for h in self._hooks]
            total = losses[0]  # `sum` would start from 0 + loss: one more kernel for the same value
            for extra in losses[1:]:
                total = total + extra
            return total
        return torch.tensor(0, dtype=torch.float, device=self._device)

    # ---- new_pretrain.py:91-96
    def _forward_pass(self, unlabeled_image, unlabeled_image_tf):
        n_unl = len(unlabeled_image)
        predict_logits = self._model(_cat_views(unlabeled_image, unlabeled_image_tf), until=self._inference_until)
        unlabeled_logits, unlabeled_tf_logits = torch.split(predict_logits, [n_unl, n_unl], dim=0)
        return unlabeled_logits, unlabeled_tf_logits

    def run(self):
        with self.meters.focus_on(self.meter_focus):
            self.meters["lr"].add([g["lr"] for g in self._optimizer.param_groups])
            self._model.train()
            try:
                self._run_pretrain()
            finally:
                _sg.gc_release(final=False)  # (the capture froze the collector's view of the heap; the trainer undoes it)
        self.close_hooks()
        return self.meters.statistics()

    def _run_pretrain(self):
        """the batch loop of ``_PretrainEpocherMixin._run_pretrain`` (new_pretrain.py:52-89): ``num_batches`` iterations
        over the (infinite) contrastive loader; the per-batch seed is drawn in ``step_compute`` (:54)."""
        for self.cur_batch_num, data in zip(range(self._num_batches), self._chain_dataloader):
            self.step(data)

    def step(self, data, seed=None):
        """One iteration of new_pretrain.py:53-89; returns the (device) regularisation loss.  Replayed from a hipGraph
        once the step's shape has been seen (``graph=`` / SPCL_STEP_GRAPH, stepgraph.py); the replay returns the captured
        step's loss tensor, which then holds the new step's value."""
        key = self._graph_key(data) if self._graph_on else None
        if key is not None:
            return self._step_staged(data, seed, key)

        def eager():
            reg_loss = self.step_compute(data, seed)
            self.step_exchange()
            self.step_update(reg_loss)
            return reg_loss
        if self._graph_on and self._device.type == "cuda":
            # (an epocher that graphs its steps keeps every backward pass on the graphs' stream: stepgraph.side_stream)
            return _sg.run_on_side_stream(eager, self._device)
        return eager()

    # ---- the captured step
    def _graph_key(self, data):
        """everything a capture of this step bakes into its launches, or None when the step cannot be captured: images'
        shape and dtype, the batch size, and what every hook declares (``EpocherHook.graph_key``: class, weight, age
        parameter ...)."""
        from ...optim import FusedRAdam
        flat = self._flat_params
        if (self._device.type != "cuda" or flat is None or not isinstance(self._optimizer, FusedRAdam)
                or (getattr(flat, "_early_idx", None) is not None and not _ddp.is_distributed()) or not self._hooks
                or tuple(self._affine_transformer._axis) != (1, 2)):
            return None
        (image, image_tf, *_), _, (partition_list, group_list) = data
        if not (torch.is_tensor(image) and torch.is_tensor(image_tf) and image.dim() == 4
                and image.shape == image_tf.shape and image.dtype == image_tf.dtype and image.is_floating_point()
                and len(partition_list) == len(image) == len(group_list)):
            return None
        if self._pair is not None and (self._pair.shape[1:] != image.shape[1:] or self._pair.shape[0] != 2 * len(image)
                                       or self._pair.dtype != image.dtype):
            return None  # a ragged last batch: that step runs eagerly, the graph of the full shape stays valid
        keys = []
        for h in self._hooks:
            k = h.graph_key() if hasattr(h, "graph_key") else None
            if k is None:
                return None
            keys.append(k)
        # (two-bucket overlap in a distributed job: the early collective starts between two compute graphs, StepGraph.cut)
        return (tuple(image.shape), image.dtype, self._inference_until, _ddp.is_distributed(),
                getattr(flat, "_early_idx", None), tuple(keys))

    def _flip_flags(self, batch):
        """the flag bytes ``TensorRandomFlip`` would act on under ``FixRandomSeed(seed)`` (new_pretrain.py:57-58)"""
        with FixRandomSeed(batch["seed"]):
            dec = self._affine_transformer.decisions(batch["n"])
        flags = [int(d[0]) | (int(d[1]) << 1) for d in dec]
        return flags + [0] * (-len(flags) % 4)

    def _step_staged(self, data, seed, key):
        seed = random.randint(0, int(1e7)) if seed is None else seed
        (image, image_tf), _, filename, unl_partition, unl_group = unzip_twice_transformed(data, self._device)
        n = len(image)
        if self.stage is None:
            self.stage = _sg.StepStage(self._device)
            self._pair = torch.empty((2 * n,) + tuple(image.shape[1:]), dtype=image.dtype, device=self._device)
            split = _ddp.is_distributed() and not _sg.collective_in_graph()
            self._step_graph = _sg.StepGraph(self._compute_staged, self.step_exchange, self.step_update, split=split)
            if split:
                self._flat_params.cutter = self._step_graph.cut
        batch = {"seed": seed, "n": n, "partition_group": list(unl_partition), "label_group": list(unl_group),
                 "filename": filename}
        # every bound slot (labels, flags) refilled from this batch; the block travels in the flip launch below when it fits
        # its kernel arguments (one eager launch in front of the replay instead of two), else by spcl_stage_bytes
        flag_off = self.stage.offset_of("flip_flags")
        fused = flag_off is not None and 0 < self.stage._used <= 3584
        self.stage.begin(batch, upload=not fused)
        try:
            flags = self.stage.bind("flip_flags", (n + 3) // 4 * 4, "u8", self._flip_flags)
            a, b = image.contiguous(), image_tf.contiguous()
            N, C, H, W = a.shape
            if fused:
                import ctypes
                dst, host, used = self.stage.host_block()
                _n.call("spcl_flip_pair_stage", _n.ptr(a), _n.ptr(b), _n.ptr(self._pair), a.element_size(), N, C, H, W,
                        ctypes.c_void_p(dst), host.ctypes.data_as(ctypes.c_void_p), used, flag_off, _n.stream())
            else:
                _n.call("spcl_flip_pair", _n.ptr(a), _n.ptr(b), _n.ptr(self._pair), a.element_size(), N, C, H, W,
                        _n.ptr(flags), _n.stream())  # the loader's tensors -> the persistent pair (eager, before the replay)
            self._staged = batch
            if hasattr(self._optimizer, "sync_lr"):
                self._optimizer.sync_lr()
            loss = self._step_graph.run(key)
            if self._step_graph.captured:
                for h in self._hooks:
                    h.after_replay()
        except BaseException:
            # ``stage.begin`` advanced the optimizer's host mirror of the step count; if the update launch did not follow
            # (a criterion's check raised, a hook aborted the step) the mirror is ahead of the device counter: drop it, the
            # next staged step reads the device's count back
            if hasattr(self._optimizer, "forget_staged_steps"):
                self._optimizer.forget_staged_steps()
            raise
        finally:
            self.stage.end()
        return loss

    def _compute_staged(self):
        batch, n = self._staged, self._staged["n"]
        _meters.begin_batch()
        return self._compute_views(self._pair[:n], self._pair[n:], batch["seed"], batch["partition_group"],
                                   batch["label_group"], batch["filename"])

    # the three phases of a step, separately callable so that a driver can capture the compute and the update in
    # hipGraphs and keep the collective outside when a whole-step capture is not possible
    def step_compute(self, data, seed=None):
        """forward + loss + backward + gradients gathered into the flat bucket (no communication, no update)."""
        seed = random.randint(0, int(1e7)) if seed is None else seed
        _meters.begin_batch()  # the step's meter adds become one launch in step_update
        (unlabeled_image, unlabeled_image_tf), _, unlabeled_filename, unl_partition, unl_group = \
            unzip_twice_transformed(data, self._device)
        if unlabeled_image.is_cuda and unlabeled_image.shape == unlabeled_image_tf.shape:
            # both views land in ONE [2n,C,H,W] buffer in one launch (view 1 copied, view 2 flipped into its half), so the
            # torch.cat of new_pretrain.py:93 is a zero-copy view in `_forward_pass`
            n_unl = len(unlabeled_image)
            with FixRandomSeed(seed):
                pair = self._affine_transformer.apply_pair(unlabeled_image,
                                                           unlabeled_image_tf.to(unlabeled_image.dtype))
            unlabeled_image, unlabeled_image_tf = pair[:n_unl], pair[n_unl:]
        else:
            with FixRandomSeed(seed):
                unlabeled_image_tf = self._affine_transformer.apply_batch(unlabeled_image_tf)
        return self._compute_views(unlabeled_image, unlabeled_image_tf, seed, unl_partition, unl_group,
                                   unlabeled_filename)

    def _compute_views(self, unlabeled_image, unlabeled_image_tf, seed, unl_partition, unl_group, unlabeled_filename):
        """new_pretrain.py:60-83 from the two (flipped) views on: forward, hooks, backward, gradient gather"""
        unlabeled_logits, unlabeled_tf_logits = self.forward_pass(unlabeled_image=unlabeled_image,
                                                                  unlabeled_image_tf=unlabeled_image_tf)
        # new_pretrain.py:64-65 flips the Conv5 "logits" as well; the InfoNCE hook only takes len() of them
        unlabeled_logits_tf = unlabeled_logits
        reg_loss = self.regularization(
            unlabeled_tf_logits=unlabeled_tf_logits, unlabeled_logits_tf=unlabeled_logits_tf, seed=seed,
            unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf, label_group=unl_group,
            partition_group=unl_partition, unlabeled_filename=unlabeled_filename,
            affine_transformer=self._affine_transformer)
        total_loss = reg_loss
        if self._flat_params is not None:
            self._flat_params.zero_grad()  # also arms the gradient sinks: backward writes into the flat bucket
            total_loss.backward(gradient=self._unit_grad(total_loss))
            self._flat_params.gather_grads()
        else:
            self._optimizer.zero_grad(set_to_none=True)
            if self._grad_bucket is not None:
                self._grad_bucket.arm_sinks()
            total_loss.backward(gradient=self._unit_grad(total_loss))
        return reg_loss

    def _unit_grad(self, loss):
        """d loss / d loss = 1 as a cached device scalar (``backward()`` would fill a fresh one every step)"""
        key = (loss.device, loss.dtype)
        one = self._ones.get(key)
        if one is None:
            one = self._ones[key] = F_hip.register_unit_gradient(torch.ones((), dtype=loss.dtype, device=loss.device))
        return one

    def step_exchange(self):
        """the step's one collective: mean of the flat gradient bucket over the ranks."""
        if self._flat_params is not None:
            self._flat_params.allreduce_()
        elif self._grad_bucket is not None:
            self._grad_bucket.allreduce()

    def step_update(self, reg_loss):
        if self.on_master():
            self.meters["reg_loss"].add(reg_loss.detach())
        from ...optim import FusedRAdam
        if isinstance(self._optimizer, FusedRAdam):  # the meters' device adds ride in the optimizer's coefficient launch
            scale = self._flat_params.grad_scale if self._flat_params is not None else 1.0  # (1 / world when fold_mean)
            self._optimizer.step(scalar_adds=_meters.take_batch(), grad_scale=scale, stage=self.stage)
        else:
            self._optimizer.step()
        _meters.flush_batch()


class PretrainDecoderEpocher(PretrainEncoderEpocher):
    """``PretrainDecoderEpocher`` (new_pretrain.py:117-126): the same loop run up to a decoder feature (SURVEY row N3;
    the only difference in the reference is the assertion on the loaders' transform freedom)."""
