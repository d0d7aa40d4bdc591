"""Mirror of the fine-tune / evaluation epochers: ``EvalEpocher`` (semi_seg/epochers/new_epocher.py:56-97) and
``FineTuneEpocher`` (:241-289; the labelled-only branch of ``SemiSupervisedEpocher``).  Same control flow and meters
(``sup_loss``, ``sup_dice`` / ``loss``, ``dice``); the network, softmax, KL_div, arg-max and Dice counts are HIP
kernels and the meters take device values (no per-step ``.item()``)."""
import os
from typing import Iterable, Optional

import torch
from torch import nn

from ... import ddp as _ddp
from ... import functional as F_hip
from ... import native as _n
from ... import stepgraph as _sg
from ...contrastyou.losses.kl import KL_div, class2one_hot
from ...contrastyou import meters as _meters
from ...contrastyou.meters import AverageValueMeter, MeterInterface, UniversalDice

_FUSED_SUP_LOSS = os.environ.get("SPCL_FUSED_SUP_LOSS", "1") != "0"  # A/B switch: 0 = the seven separate launches


def unzip_single_transformed(data, device):
    """``preprocess_input_with_single_transformation`` (semi_seg/epochers/helper.py:39-45): ((image, target), filename,
    (partition, group)) -> image, target, filename, partition, group."""
    (image, target), filename, (partition_list, group_list) = data
    return (image.to(device, non_blocking=True), target.to(device, non_blocking=True), filename, partition_list,
            group_list)


def unzip_twice_transformed_labeled(data, device):
    """``preprocess_input_with_twice_transformation`` for a labelled batch: ((image, image_tf, target, target_tf), ...)."""
    (image, image_tf, target, target_tf), filename, (partition_list, group_list) = data
    return ((image.to(device, non_blocking=True), image_tf.to(device, non_blocking=True)),
            target.to(device, non_blocking=True), filename, partition_list, group_list)


class _EpocherBase:
    meter_focus = "tra"

    def __init__(self, *, model: nn.Module, num_batches: int, cur_epoch=0, device="cuda"):
        self._model, self._num_batches, self._cur_epoch = model, num_batches, cur_epoch
        self._device = torch.device(device)
        self.meters = MeterInterface(default_focus=self.meter_focus)
        with self.meters.focus_on(self.meter_focus):
            self.configure_meters(self.meters)
        self.cur_batch_num = 0

    @property
    def num_classes(self):
        return self._model.num_classes

    def configure_meters(self, meters):
        return meters

    def init(self):
        pass

    @staticmethod
    def on_master():
        return _ddp.on_master()

    def run(self):
        with self.meters.focus_on(self.meter_focus):
            try:
                self._run()
            finally:
                _sg.gc_release(final=False)  # (a capture froze the collector's view of the heap; the trainer undoes it)
        return self.meters.statistics()


class _EvalGraphs:
    """hipGraphs of one model's validation batch, one per batch shape (the reference's validation loaders hand over one SCAN
    per batch, ``ScanBatchSampler``: a dozen distinct slice counts), kept ON THE MODEL across the EvalEpochers of a training
    run (the trainer builds two per epoch, val + test).  A validation batch is ~70 launches of 3 - 10 us each; issued one by
    one the pass is bound by the host (1.58 ms per batch at 8 and at 32 slices, tools/diag/eval_speed.py) -- and the
    reference's fine-tune epoch is 200 training steps against ~200 validation batches.

    An entry replays: forward in ``eval()`` mode, softmax, one-hot, criterion, arg-max, Dice counts -- the very launches of
    the eager pass, so the values are the eager pass's bit for bit.  What a capture bakes in is the entry's key: shapes,
    dtypes, criterion, and the storage of every parameter and buffer (``signature``: a FlatParams built later MOVES the
    parameters; entries of another signature are dropped)."""

    def __init__(self):
        self.signature = None
        self.entries = {}   # key -> dict(seen=, graph=, img=, tgt=, out=)
        self.pool = None

    @staticmethod
    def of(model):
        g = model.__dict__.get("_spcl_eval_graphs")
        if g is None:
            g = model.__dict__["_spcl_eval_graphs"] = _EvalGraphs()
        return g

    @staticmethod
    def signature_of(model):
        return hash(tuple(t.data_ptr() for t in model.parameters()) + tuple(t.data_ptr() for t in model.buffers())
                    + (str(getattr(model, "_compute_dtype", None)),))

    def sync(self, model):
        sig = self.signature_of(model)
        if sig != self.signature:
            self.signature, self.entries, self.pool = sig, {}, None


class EvalEpocher(_EpocherBase):
    meter_focus = "eval"

    def __init__(self, *, model: nn.Module, loader: Iterable, sup_criterion, cur_epoch=0, device="cuda",
                 graph: Optional[bool] = None):
        self._loader = loader
        self._sup_criterion = sup_criterion
        super().__init__(model=model, num_batches=len(loader), cur_epoch=cur_epoch, device=device)
        self._graph_on = (_sg.graph_default() if graph is None else bool(graph)) and self._device.type == "cuda"

    def configure_meters(self, meters):
        C = self.num_classes
        meters.register_meter("loss", AverageValueMeter())
        meters.register_meter("dice", UniversalDice(C, report_axises=list(range(1, C))))
        return meters

    def get_score(self):
        with self.meters.focus_on(self.meter_focus):
            return self.meters["dice"].summary()["DSC_mean"]

    def _run(self):
        self._model.eval()
        return self._run_eval()

    def _batch(self, eval_img, eval_target):
        """one validation batch (new_epocher.py:84-90) -> (loss, per-sample intersections, unions)"""
        eval_logits = self._model(eval_img)
        onehot_target = class2one_hot(eval_target.squeeze(1), self.num_classes)
        eval_loss = self._sup_criterion(F_hip.softmax_classes(eval_logits), onehot_target, disable_assert=True)
        inter, union = F_hip.dice_counts(F_hip.argmax_classes(eval_logits), eval_target.squeeze(1), self.num_classes)
        return eval_loss, inter, union

    def _batch_replayed(self, graphs, eval_img, eval_target):
        """``_batch`` from a hipGraph of this shape: first sight eager (lazily created workspaces are then outside any
        graph's pool), second sight captured, from then on copy-in + replay + copy-out"""
        key = (tuple(eval_img.shape), eval_img.dtype, tuple(eval_target.shape), eval_target.dtype,
               type(self._sup_criterion).__name__, getattr(self._sup_criterion, "_eps", None), self.num_classes)
        e = graphs.entries.get(key)
        if e is None:
            graphs.entries[key] = {"graph": None}
            return _sg.run_on_side_stream(lambda: self._batch(eval_img, eval_target), self._device)
        if e["graph"] is None:
            if e.get("failed"):
                return _sg.run_on_side_stream(lambda: self._batch(eval_img, eval_target), self._device)
            e["img"], e["tgt"] = torch.empty_like(eval_img), torch.empty_like(eval_target)
            g = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(g, pool=graphs.pool, stream=_sg.side_stream(self._device)):
                    e["out"] = self._batch(e["img"], e["tgt"])
            except Exception as exc:  # noqa: BLE001 -- a capture that fails must not end the validation pass
                import warnings
                e["failed"] = True
                try:
                    torch.cuda.synchronize()
                except Exception:  # noqa: BLE001
                    pass
                warnings.warn(f"hipGraph capture of the validation batch failed ({type(exc).__name__}: {exc}); this shape "
                              "continues with eager launches")
                return _sg.run_on_side_stream(lambda: self._batch(eval_img, eval_target), self._device)
            e["graph"] = g
            if graphs.pool is None:
                graphs.pool = g.pool()  # one pool for every shape: the batches replay one after the other
        e["img"].copy_(eval_img, non_blocking=True)
        e["tgt"].copy_(eval_target, non_blocking=True)
        e["graph"].replay()
        loss, inter, union = e["out"]  # (rewritten by the next replay of this shape: the meters get copies)
        return loss.clone(), inter.clone(), union.clone()

    @torch.no_grad()
    def _run_eval(self):
        graphs = None
        if self._graph_on:
            graphs = _EvalGraphs.of(self._model)
            graphs.sync(self._model)
        for self.cur_batch_num, eval_data in zip(range(self._num_batches), self._loader):
            eval_img, eval_target, file_path, _, group = unzip_single_transformed(eval_data, self._device)
            if graphs is not None and eval_img.is_cuda and eval_img.is_contiguous() and eval_target.is_contiguous():
                eval_loss, inter, union = self._batch_replayed(graphs, eval_img, eval_target)
            else:
                eval_loss, inter, union = self._batch(eval_img, eval_target)
            self.meters["loss"].add(eval_loss)
            dice = self.meters["dice"]
            dice.add_counts(inter, union, dice.group_names_for(inter.shape[0], list(group)))


class FineTuneEpocher(_EpocherBase):
    meter_focus = "semi"

    def __init__(self, *, model: nn.Module, optimizer, labeled_loader: Iterable, sup_criterion, num_batches: int,
                 cur_epoch=0, device="cuda", flat_params: Optional[_ddp.FlatParams] = None,
                 graph: Optional[bool] = None, **kwargs):
        self._optimizer = optimizer
        self._labeled_loader = labeled_loader
        self._sup_criterion = sup_criterion
        self._flat_params = flat_params
        from ...optim import FusedRAdam
        if flat_params is not None:
            # FusedRAdam: the exchange leaves the ranks' SUM, 1 / world is applied inside the RAdam kernel; any other
            # optimizer reads the bucket as it is and must find the MEAN there (the flag is state of the shared FlatParams:
            # an earlier epocher may have set it)
            flat_params.fold_mean = isinstance(optimizer, FusedRAdam)
        self._unit = None
        # the step as a hipGraph (stepgraph.py): image and label map are copied into persistent buffers in front of the
        # replay; the Dice counts come back in persistent [B, C] tensors and are handed to the meter after it
        self._graph_on = _sg.graph_default() if graph is None else bool(graph)
        self._step_graph = None
        self._static = None  # (image, target) buffers of the captured step
        self._counts = None
        super().__init__(model=model, num_batches=num_batches, cur_epoch=cur_epoch, device=device)

    def configure_meters(self, meters):
        C = self.num_classes
        meters.register_meter("lr", AverageValueMeter())
        meters.register_meter("sup_loss", AverageValueMeter())
        meters.register_meter("sup_dice", UniversalDice(C, report_axises=list(range(1, C))))
        return meters

    def _run(self):
        self.meters["lr"].add([g["lr"] for g in self._optimizer.param_groups])
        self._model.train()
        return self._run_only_label()

    def _forward_pass(self, labeled_image):
        return self._model(labeled_image)

    def step(self, labeled_data):
        """one iteration of ``_run_only_label`` (new_epocher.py:260-283); returns the (device) supervised loss.  Replayed
        from a hipGraph once the batch shape has been seen (``graph=`` / SPCL_STEP_GRAPH)."""
        (labeled_image, _), labeled_target, labeled_filename, _, label_group = \
            unzip_twice_transformed_labeled(labeled_data, self._device)
        key = self._graph_key(labeled_image, labeled_target) if self._graph_on else None
        if key is None:
            def eager():
                loss = self.step_compute(labeled_image, labeled_target)
                self.step_exchange()
                self.step_update(loss)
                return loss
            if self._graph_on and self._device.type == "cuda":  # (keep every backward pass on the graphs' stream)
                sup_loss = _sg.run_on_side_stream(eager, self._device)
            else:
                sup_loss = eager()
            inter, union = self._counts
        else:
            if self._static is None:
                self._static = (torch.empty_like(labeled_image), torch.empty_like(labeled_target))
                self._step_graph = _sg.StepGraph(lambda: self.step_compute(*self._static), self.step_exchange,
                                                 self.step_update, split=_ddp.is_distributed())
            si, stg = self._static
            a, b = labeled_image, labeled_target
            if (a.is_contiguous() and b.is_contiguous() and a.dtype == si.dtype and b.dtype == stg.dtype
                    and a.device == si.device and b.device == stg.device
                    and (a.numel() * a.element_size()) % 16 == 0 and (b.numel() * b.element_size()) % 16 == 0
                    and not ((a.data_ptr() | b.data_ptr() | si.data_ptr() | stg.data_ptr()) % 16)):
                _n.call("spcl_copy_pair", _n.ptr(si), _n.ptr(a), a.numel() * a.element_size(), _n.ptr(stg), _n.ptr(b),
                        b.numel() * b.element_size(), _n.stream())  # (both copies in one launch)
            else:
                si.copy_(a, non_blocking=True)
                stg.copy_(b, non_blocking=True)
            if hasattr(self._optimizer, "sync_lr"):
                self._optimizer.sync_lr()
            sup_loss = self._step_graph.run(key)
            inter, union = self._counts  # the captured step's result tensors are rewritten by the next replay: keep copies
            if inter._base is not None and inter._base is union._base:
                inter, union = inter._base.clone().unbind(0)  # (the fused criterion's two halves of one buffer: one launch)
            else:
                inter, union = inter.clone(), union.clone()
        if self.on_master():
            dice = self.meters["sup_dice"]
            dice.add_counts(inter, union, dice.group_names_for(inter.shape[0], list(label_group)))
        return sup_loss

    def _graph_key(self, image, target):
        from ...optim import FusedRAdam
        if (self._device.type != "cuda" or self._flat_params is None or not isinstance(self._optimizer, FusedRAdam)
                or getattr(self._flat_params, "_early_idx", None) is not None):
            return None
        if self._static is not None and (self._static[0].shape != image.shape or self._static[0].dtype != image.dtype
                                         or self._static[1].shape != target.shape
                                         or self._static[1].dtype != target.dtype):
            return None  # ragged last batch: eager
        return (tuple(image.shape), image.dtype, tuple(target.shape), target.dtype, _ddp.is_distributed(),
                type(self._sup_criterion).__name__)

    # the three phases of a step (compute / collective / update), as in the pre-train epocher
    def step_compute(self, labeled_image, labeled_target):
        label_logits = self._forward_pass(labeled_image)
        fused = (_FUSED_SUP_LOSS and isinstance(self._sup_criterion, KL_div) and label_logits.is_cuda
                 and label_logits.shape[1] == self.num_classes <= 16 and label_logits.shape[0] <= 1024)
        if fused:
            # new_epocher.py:268-282 in one launch: softmax, one-hot, KL_div, arg-max and the Dice counts (and, for the unit
            # gradient the loop backpropagates, the gradient w.r.t. the logits): functional._SupLossFn
            sup_loss, counts = F_hip.sup_loss_kl_onehot(label_logits, labeled_target.squeeze(1), self._sup_criterion._eps)
        else:
            onehot_target = class2one_hot(labeled_target.squeeze(1), self.num_classes)
            sup_loss = self._sup_criterion(F_hip.softmax_classes(label_logits), onehot_target, disable_assert=True)
        if self._unit is None or self._unit.device != sup_loss.device or self._unit.dtype != sup_loss.dtype:
            self._unit = F_hip.register_unit_gradient(torch.ones((), dtype=sup_loss.dtype, device=sup_loss.device))
        if self._flat_params is not None:
            self._flat_params.zero_grad()  # arms the gradient sinks: backward fills the flat bucket in place
            sup_loss.backward(gradient=self._unit)
            self._flat_params.gather_grads()
        else:
            self._optimizer.zero_grad(set_to_none=True)
            sup_loss.backward(gradient=self._unit)
        if fused:
            self._counts = counts
        else:
            with torch.no_grad():  # Dice counts of the training batch (new_epocher.py:279-282): [B, C] intersections / unions
                self._counts = F_hip.dice_counts(F_hip.argmax_classes(label_logits.detach()), labeled_target.squeeze(1),
                                                 self.num_classes)
        return sup_loss

    def step_exchange(self):
        if self._flat_params is not None:
            self._flat_params.allreduce_()

    def step_update(self, sup_loss):
        from ...optim import FusedRAdam
        if isinstance(self._optimizer, FusedRAdam) and self._flat_params is not None:
            adds = None
            if self.on_master():  # the meter's device add rides in the optimizer's coefficient launch (as in pre-training)
                _meters.begin_batch()
                self.meters["sup_loss"].add(sup_loss.detach())
                adds = _meters.take_batch()
            self._optimizer.step(scalar_adds=adds, grad_scale=self._flat_params.grad_scale)  # (1 / world: the exchange left the sum)
            _meters.flush_batch()
        else:
            self._optimizer.step()
            if self.on_master():
                with torch.no_grad():
                    _meters.begin_batch()
                    self.meters["sup_loss"].add(sup_loss.detach())
                    _meters.flush_batch()

    def _run_only_label(self):
        for self.cur_batch_num, labeled_data in zip(range(self._num_batches), self._labeled_loader):
            self.step(labeled_data)
