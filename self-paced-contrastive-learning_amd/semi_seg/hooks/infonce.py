"""Mirror of ``semi_seg/hooks/infonce.py`` for the encoder pre-train path: ``PScheduler`` (:34-53), ``INFONCEHook``
(:56-110), ``SelfPacedINFONCEHook`` (:113-141), ``_INFONCEEpochHook`` (:144-198), ``_SPINFONCEEpochHook`` (:244-268).
The projector and criterion are the HIP-backed mirrors; meters take device scalars (no per-step ``.item()``).
Dense/decoder hooks (``_INFONCEDenseHook`` :201-241) are SURVEY row N3 and are not built.

Differences that do not change results: the TensorBoard figure taps of the first batches (:185-192,264-266) are
delivered to an optional ``tap_callback(name, tensor, epocher)`` instead of a global writer; the feature flip before the
projector (:177-178) is skipped when the projector pools to (1,1) -- a global average is flip-invariant."""
import math
import os
from typing import List

import torch
from torch import nn

from ... import functional as F_hip
from ...contrastyou.hooks.base import EpocherHook, TrainerHook
from ...contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss, SupConLoss1, supcon_heads
from ...contrastyou.meters import AverageValueMeter
from ..arch.hook import SingleFeatureExtractor
from ..arch.unet import UNet
from ..epochers.helper import FixRandomSeed
from .utils import get_label, meter_focus

_FUSE_NORM = os.environ.get("SPCL_FUSE_NORM", "1") != "0"  # A/B switch: 0 keeps the projector's normalisation launches

encoder_names = list(UNet.encoder_names)
decoder_names = list(UNet.decoder_names)


class PScheduler:
    """age-parameter schedule of the self-paced loss: ``begin + (end - begin) * (epoch / max_epoch) ** p``; ``step()``
    once per epoch (reference :34-53, a deepclustering2 ``WeightScheduler``)."""

    def __init__(self, max_epoch, begin_value=0.0, end_value=1.0, p=0.5):
        self.max_epoch, self.p, self.epoch = max_epoch, p, 0
        self.begin_value, self.end_value = float(begin_value), float(end_value)

    def get_lr(self, cur_epoch):
        progress = math.pow(cur_epoch / self.max_epoch, self.p)
        return self.begin_value + (self.end_value - self.begin_value) * progress

    value = property(lambda self: self.get_lr(self.epoch))

    def step(self):
        self.epoch += 1

    def state_dict(self):
        return {"epoch": self.epoch}

    def load_state_dict(self, sd):
        self.epoch = sd["epoch"]


class INFONCEHook(TrainerHook):
    """Trainer-side InfoNCE hook on one encoder feature: owns the feature tap, the projector (learnable) and the
    criterion; each call (one per epoch) hands out the epocher hook that computes the loss."""
    _epoch_hook_class = None  # set below, once the epocher hooks exist

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, spatial_size=None,
                 data_name: str, contrast_on: str, sync_checks: bool = True, tap_callback=None) -> None:
        super().__init__(hook_name=name)
        assert feature_name in encoder_names + decoder_names, feature_name
        self._feature_name, self._weight = feature_name, weight
        self._sync_checks, self._tap_callback = sync_checks, tap_callback
        self._contrast_on, self._data_name = contrast_on, data_name
        self._extractor = SingleFeatureExtractor(model, feature_name=feature_name)
        if feature_name in encoder_names:  # :73-76
            spatial_size = spatial_size or (1, 1)
        else:
            spatial_size = spatial_size or (10, 10)
        self._projector = self.init_projector(input_dim=model.get_channel_dim(feature_name), spatial_size=spatial_size)
        self._criterion = self.init_criterion()
        self._learnable_models = (self._projector,)

    # -- pieces a subclass may swap
    @property
    def projector_class(self):  # :101-106
        from ...contrastyou.projectors.heads import DenseProjectionHead, ProjectionHead
        return ProjectionHead if self.is_encoder else DenseProjectionHead

    def init_projector(self, *, input_dim, spatial_size):
        return self.projector_class(input_dim=input_dim, hidden_dim=256, output_dim=256, head_type="mlp",
                                    normalize=True, spatial_size=spatial_size)

    def init_criterion(self):
        self._criterion = SupConLoss1(sync_checks=self._sync_checks)
        return self._criterion

    # -- TrainerHook interface
    @property
    def learnable_modules(self) -> List[nn.Module]:
        return [self._projector]

    @property
    def is_encoder(self):
        return self._feature_name in encoder_names

    def _label_generator(self, *, partition_group, label_group):
        return get_label(self._contrast_on, self._data_name, partition_group, label_group)

    def _new_epoch_hook(self):
        cls = self._epoch_hook_class if self.is_encoder else self._dense_hook_class  # :83-91
        return cls(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                      projector=self._projector, criterion=self._criterion,
                                      label_generator=self._label_generator, tap_callback=self._tap_callback)

    def __call__(self):
        return self._new_epoch_hook()


class SelfPacedINFONCEHook(INFONCEHook):
    """INFONCEHook with the self-paced criterion; every epoch it moves the age parameter gamma along the PScheduler"""

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, spatial_size=(1, 1),
                 data_name: str, contrast_on: str, mode="soft", p=0.5, begin_value=1e6, end_value=1e6,
                 correct_grad: bool = False, max_epoch: int, sync_checks: bool = True, tap_callback=None) -> None:
        if feature_name not in encoder_names:
            # the reference constructs this combination and then fails in the first batch: its __call__ (:133-141) always
            # hands out the encoder-style epoch hook, which feeds the dense head's [2B, C, h, w] map to the criterion
            raise NotImplementedError("SelfPacedINFONCEHook on a decoder feature: the reference has no dense self-paced "
                                      "hook (semi_seg/hooks/infonce.py:133-141); use INFONCEHook")
        # needed by init_criterion, which the base constructor calls
        self._mode, self._correct_grad = mode, correct_grad
        self._p, self._max_epoch = float(p), int(max_epoch)
        self._begin_value, self._end_value = float(begin_value), float(end_value)
        super().__init__(name=name, model=model, feature_name=feature_name, weight=weight, spatial_size=spatial_size,
                         data_name=data_name, contrast_on=contrast_on, sync_checks=sync_checks,
                         tap_callback=tap_callback)

    def init_criterion(self):
        self._scheduler = PScheduler(max_epoch=self._max_epoch, begin_value=self._begin_value,
                                     end_value=self._end_value, p=self._p)
        self._criterion = SelfPacedSupConLoss(weight_update=self._mode, correct_grad=self._correct_grad,
                                              sync_checks=self._sync_checks)
        return self._criterion

    def __call__(self):
        self._criterion.set_gamma(self._scheduler.value)  # this epoch's gamma, then advance the schedule
        self._scheduler.step()
        return self._new_epoch_hook()


class _INFONCEEpochHook(EpocherHook):
    """per-epoch InfoNCE hook: taps the feature during the forward pass, then projector -> criterion on the two views"""
    _taps_first_batch = ("pos_mask", "sim_exp", "sim_logits")

    def __init__(self, *, name: str, weight: float, extractor, projector, criterion, label_generator,
                 tap_callback=None) -> None:
        super().__init__(name)
        self._weight, self._projector, self._criterion = weight, projector, criterion
        self._label_generator, self._tap_callback = label_generator, tap_callback
        self._extractor = extractor
        self._extractor.bind()
        self._n = 0             # batches seen this epoch
        self._label_cache = {}  # batch composition -> device label tensor
        self._share_pool = False  # set by CombineEpochHook when several hooks pool the same feature
        self._batch_group = None  # ... and the hooks whose heads then run as one batched projection

    @property
    def shared_pool_key(self):
        """hooks with equal keys average-pool the same tensor to (1, 1): they can share the pooled rows"""
        p = self._projector
        if type(p).__name__ != "ProjectionHead" or tuple(getattr(p, "_spatial_size", ())) != (1, 1) \
                or getattr(p, "_pool_name", None) != "adaptive_avg":
            return None
        return (id(self._extractor._model), self._extractor._feature_name)

    @staticmethod
    def batchable_heads(members):
        """2..4 hooks whose ProjectionHeads are MLP heads of one shape (the K meta-label hooks of
        hooks/creator.py:102-124 all are): functional.projector_heads runs them together"""
        ps = [m._projector for m in members]
        if not 2 <= len(ps) <= 4 or any(getattr(p, "_head_type", None) != "mlp" for p in ps):
            return False
        shape = lambda p: (tuple(p._header[2].weight.shape), tuple(p._header[4].weight.shape), bool(p._normalize))  # noqa: E731
        return all(shape(p) == shape(ps[0]) for p in ps)

    def _fuse_norm(self):
        """may the projector's F.normalize run inside the criterion's launch?  (ProjectionHead(normalize=True) in front of
        one of the two supervised-contrastive criteria: ``projector(x, normalize=False)`` + ``criterion(...,
        normalize_inputs=True)`` is the same function of the feature, two launches shorter)"""
        if not _FUSE_NORM:
            return False
        members = self._batch_group if self._batch_group is not None else (self,)
        return all(type(m._projector).__name__ == "ProjectionHead" and bool(getattr(m._projector, "_normalize", False))
                   and isinstance(m._criterion, (SupConLoss1, SelfPacedSupConLoss)) for m in members)

    def _project(self, feature, base, raw=False):
        """z of this hook's head (``raw``: its rows before F.normalize); with a batch group the first hook of the step
        projects for all of them (cached on the tapped tensor, which lives exactly one step)"""
        if self._batch_group is None:
            return self._projector(feature, normalize=False) if raw else self._projector(feature)
        cache = getattr(base, "_spcl_zs", None)
        if cache is None or cache[0] != (feature.shape[0], raw):
            heads = [(p._header[2].weight, p._header[2].bias, p._header[4].weight, p._header[4].bias)
                     for p in (m._projector for m in self._batch_group)]
            zs = F_hip.projector_heads(feature, heads, self._projector._normalize and not raw)
            cache = ((feature.shape[0], raw), {id(m._projector): z for m, z in zip(self._batch_group, zs)})
            try:
                base._spcl_zs = cache
            except AttributeError:
                pass
        return cache[1][id(self._projector)]

    def _group_loss(self, feature, base, partition_group, label_group, raw=False):
        """This hook's loss out of the group's batched evaluation -- the first hook of the step projects for all heads AND
        evaluates all K criteria in the launches of one (contrast_loss3.supcon_heads; row N4), cached on the tapped tensor
        like the projections -- or None when there is no group or its criteria cannot be batched."""
        if self._batch_group is None:
            return None
        cache = getattr(base, "_spcl_losses", None)
        if cache is None or cache[0] != feature.shape[0]:
            members = self._batch_group
            zs = [m._project(feature, base, raw) for m in members]
            targets = [m._labels(partition_group, label_group, feature.device) for m in members]
            losses = supcon_heads([m._criterion for m in members], zs, targets, normalize_inputs=raw)
            cache = (feature.shape[0], None if losses is None else {id(m): l for m, l in zip(members, losses)})
            try:
                base._spcl_losses = cache
            except AttributeError:
                return None  # (nowhere to keep the other heads' losses: evaluate one by one)
        return None if cache[1] is None else cache[1][id(self)]

    @meter_focus
    def configure_meters(self, meters):
        meters = super().configure_meters(meters)
        meters.register_meter("loss", AverageValueMeter())
        return meters

    def before_forward_pass(self, **kwargs):
        self._extractor.clear()
        self._extractor.set_enable(True)

    def after_forward_pass(self, **kwargs):
        self._extractor.set_enable(False)

    def _labels(self, partition_group, label_group, device):
        """labels as a device tensor; cached per distinct (partition, group) content so that a repeated batch
        composition (the synthetic benchmark) costs no host->device copy.  In a staged step (the epocher replays its
        step from a hipGraph, stepgraph.py) the vector lives in this hook's slot of the epocher's stage, which the
        epocher refills from every new batch."""
        stage = getattr(getattr(self, "_epocher", None), "stage", None)
        if stage is not None and stage.active:
            return stage.bind(("labels", id(self)), len(partition_group), "f32",
                              lambda b: self._label_generator(partition_group=b["partition_group"],
                                                              label_group=b["label_group"]))
        key = (tuple(partition_group), tuple(label_group))
        t = self._label_cache.get(key)
        if t is None:
            labels = self._label_generator(partition_group=partition_group, label_group=label_group)
            t = torch.tensor(labels, dtype=torch.float32, device=device)
            if len(self._label_cache) < 64:
                self._label_cache[key] = t
        return t

    def _two_views(self, n_unl, affine_transformer, seed):
        """[view 1 | view 2] features of the current batch as one tensor.  The reference flips view 1's features with the
        sample-wise random flip before projecting (:176-180); when the projector pools to (1, 1) a global average is
        flip-invariant, so the flip (and the stack / cat copies around it) is skipped."""
        feature = self._extractor.feature()
        if feature.shape[0] != 2 * n_unl:  # a slice costs a zero-fill + strided copy in backward: only when needed
            feature = feature[-2 * n_unl:]
        if tuple(getattr(self._projector, "_spatial_size", (1, 1))) == (1, 1):
            if self._batch_group is not None:
                return feature  # the batched projection pools it (once for all heads)
            if self._share_pool:
                # K hooks on one feature (run_self_paced_acdc:61-70 / hooks/creator.py:102-124): pool ONCE -- the K
                # projectors then read [2n, C] rows and autograd adds K [2n, C] gradients before ONE pooling backward,
                # instead of K passes over the feature map forward and K full-size gradients + K - 1 adds backward
                pooled = getattr(feature, "_spcl_pooled", None)
                if pooled is None:
                    pooled = F_hip.adaptive_pool2d(feature, (1, 1), "avg")
                    try:
                        feature._spcl_pooled = pooled  # lives as long as the tapped tensor (this step)
                    except AttributeError:
                        pass
                return pooled
            return feature
        first, second = torch.chunk(feature, 2, dim=0)
        with FixRandomSeed(seed):
            first = torch.stack([affine_transformer(x) for x in first], dim=0)
        return torch.cat([first, second], dim=0)

    def _record(self, loss):
        self.meters["loss"].add(loss.detach())
        if self._n == 0 and self._tap_callback is not None:
            for tap in self._taps_first_batch:
                self._tap_callback(tap, getattr(self._criterion, tap), self.epocher)
        self._n += 1

    @meter_focus
    def __call__(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf, partition_group,
                 label_group, **kwargs):
        feature = self._two_views(len(unlabeled_logits_tf), affine_transformer, seed)
        base = self._extractor.feature()
        raw = self._fuse_norm()
        loss = self._group_loss(feature, base, partition_group, label_group, raw)
        if loss is None:
            z_first, z_second = torch.chunk(self._project(feature, base, raw), 2)
            loss = self._criterion(z_first, z_second, target=self._labels(partition_group, label_group, feature.device),
                                   **({"normalize_inputs": True} if raw else {}))
        self._record(loss)
        return loss if self._weight == 1 else loss * self._weight

    def close(self):
        self._extractor.remove()
        if hasattr(self._criterion, "flush_check"):
            self._criterion.flush_check()  # (the last replayed step's check, see ``after_replay``)

    def graph_key(self):
        """replayable when the projector pools to (1, 1) (no seeded feature flip, `_two_views`) and no TensorBoard tap is
        still due; bakes the loss weight and the criterion's scalars (age parameter: changes per epoch)"""
        if tuple(getattr(self._projector, "_spatial_size", (1, 1))) != (1, 1):
            return None
        if self._tap_callback is not None and self._n < 2:
            return None
        c = self._criterion
        return (type(self).__name__, self._name, float(self._weight), type(c).__name__, float(c._t),
                getattr(c, "age_param", None), getattr(c, "_weight_update", None), getattr(c, "_correct_grad", None),
                getattr(c, "_exclude_pos", None))

    def after_replay(self):
        c = self._criterion
        c._taps_cache = c._host_out = None  # the captured result block now holds the new step's values
        if c.sync_checks:
            c.check_lagged()  # the reference's unit-norm assertion / NaN guard, without draining the queue (one step late)


class _SPINFONCEEpochHook(_INFONCEEpochHook):
    @meter_focus
    def configure_meters(self, meters):
        meters = super().configure_meters(meters)
        meters.register_meter("sp_weight", AverageValueMeter())
        meters.register_meter("age_param", AverageValueMeter())
        return meters

    @meter_focus
    def __call__(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf, partition_group,
                 label_group, **kwargs):
        loss = super().__call__(affine_transformer=affine_transformer, seed=seed,
                                unlabeled_tf_logits=unlabeled_tf_logits, unlabeled_logits_tf=unlabeled_logits_tf,
                                partition_group=partition_group, label_group=label_group, **kwargs)
        self.meters["sp_weight"].add(self._criterion.downgrade_ratio_tensor)
        self.meters["age_param"].add(self._criterion.age_param)
        if self._n == 1 and self._tap_callback is not None:
            self._tap_callback("sp_mask", self._criterion.sp_mask, self.epocher)
        return loss


def get_n_point_coordinate(h, w, n):
    """:20-22: n distinct rows x n distinct columns, drawn from numpy's global RNG"""
    import numpy as np
    return [(x, y) for x, y in zip(np.random.choice(range(h), n, replace=False),
                                   np.random.choice(range(w), n, replace=False))]


class _INFONCEDenseHook(_INFONCEEpochHook):
    """:201-241: contrast ``point_nums`` pixels per slice of the dense (decoder) projection between the
    two views -- every sampled point is its own class, its positive is the same point of the other view."""
    point_nums = 5

    def graph_key(self):
        """replayable (round 6): what the step draws on the host -- the sample-wise feature flips and the points of every
        slice -- reaches the captured launches through the epocher's stage (flag bytes, point indices), like the label vectors
        of the global hooks"""
        if self._tap_callback is not None and self._n < 2:
            return None
        c = self._criterion
        return (type(self).__name__, self._name, float(self._weight), type(c).__name__, float(c._t), self.point_nums,
                tuple(getattr(self._projector, "_spatial_size", ())))

    def _staged_points(self, stage, B, h, w):
        """[3][B * point_nums] int32 device view (slice, row, column of every sampled point), refilled from every new batch's
        seed exactly as ``region_extractor`` draws them (slice by slice, numpy's RNG under FixRandomSeed(seed))"""
        P = self.point_nums

        def fill(batch):
            bi, xi, yi = [], [], []
            with FixRandomSeed(batch["seed"]):
                for b in range(B):
                    for x, y in get_n_point_coordinate(n=P, h=h, w=w):
                        bi.append(b), xi.append(int(x)), yi.append(int(y))
            return bi + xi + yi
        return stage.bind(("dense_points", id(self), B, h, w), 3 * B * P, "i32", fill).view(3, B * P)

    @staticmethod
    def _gather_points(z, lin, B, P):
        """z [B, C, h, w], lin [B * P] flat pixel indices (slice b's P points in order) -> [B * P, C]: what
        ``z[bi, :, xi, yi]`` returns, as one gather along the flattened pixels (its backward is a scatter-add onto distinct
        positions: capturable, and the same values)"""
        C = z.shape[1]
        zf = z.reshape(B, C, -1)
        got = torch.gather(zf, 2, lin.view(B, 1, P).expand(B, C, P))
        return got.permute(0, 2, 1).reshape(B * P, C)

    @meter_focus
    def __call__(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf, partition_group,
                 label_group, **kwargs):
        n_unl = len(unlabeled_logits_tf)
        feature = self._extractor.feature()[-n_unl * 2:]
        first, second = torch.chunk(feature, 2, dim=0)
        stage = getattr(getattr(self, "_epocher", None), "stage", None)
        staged = stage is not None and stage.active and first.is_cuda and hasattr(self._epocher, "_flip_flags")
        if staged:
            # a staged step (the epocher replays it from a hipGraph): the flags are the epocher's own slot -- view 1's features
            # get the sample-wise flips view 2's images got, same seed, same draws (:206-207)
            flags = stage.bind("flip_flags", (n_unl + 3) // 4 * 4, "u8", self._epocher._flip_flags)
            first = F_hip.flip_batch(first, flags[:n_unl])
        else:
            with FixRandomSeed(seed):  # view 1's features get the sample-wise flips view 2's images got (:206-207)
                first = affine_transformer.apply_batch(first) if hasattr(affine_transformer, "apply_batch") else \
                    torch.stack([affine_transformer(x) for x in first], dim=0)
        z_first, z_second = torch.chunk(self._projector(torch.cat([first, second.contiguous()], dim=0)), 2)
        if staged:
            B, _, h, w = z_first.shape
            pts = self._staged_points(stage, B, h, w).long()
            lin = pts[1] * w + pts[2]
            a = self._gather_points(z_first, lin, B, self.point_nums)
            b = self._gather_points(z_second, lin, B, self.point_nums)
        else:
            with FixRandomSeed(seed):
                a = self.region_extractor(z_first, point_nums=self.point_nums)
            with FixRandomSeed(seed):
                b = self.region_extractor(z_second, point_nums=self.point_nums)
        labels = torch.arange(a.shape[0], dtype=torch.float32, device=a.device)
        loss = self._criterion(a, b, target=labels)
        self._record(loss)
        return loss if self._weight == 1 else loss * self._weight

    @staticmethod
    def region_extractor(normalize_features, point_nums=5):
        """:228-237: per slice ``point_nums`` pixels (distinct rows, distinct columns) -> [B * point_nums, C]; the draws
        happen slice by slice in the reference's order, the pixels are fetched by one gather"""
        h, w = normalize_features.shape[2:]
        bi, xi, yi = [], [], []
        for b in range(normalize_features.shape[0]):
            for x, y in get_n_point_coordinate(n=point_nums, h=h, w=w):
                bi.append(b), xi.append(int(x)), yi.append(int(y))
        dev = normalize_features.device
        return normalize_features[torch.tensor(bi, device=dev), :, torch.tensor(xi, device=dev), torch.tensor(yi, device=dev)]


INFONCEHook._epoch_hook_class = _INFONCEEpochHook
INFONCEHook._dense_hook_class = _INFONCEDenseHook
SelfPacedINFONCEHook._epoch_hook_class = _SPINFONCEEpochHook
