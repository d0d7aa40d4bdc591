"""Mirror of ``semi_seg/hooks/infonce.py`` for the encoder pre-train path: ``PScheduler`` (:34-53), ``INFONCEHook``
(:56-110), ``SelfPacedINFONCEHook`` (:113-141), ``_INFONCEEpochHook`` (:144-198), ``_SPINFONCEEpochHook`` (:244-268).
The projector and criterion are the HIP-backed mirrors; meters take device scalars (no per-step ``.item()``).
Dense/decoder hooks (``_INFONCEDenseHook`` :201-241) are SURVEY row N3 and are not built.

Differences that do not change results: the TensorBoard figure taps of the first batches (:185-192,264-266) are
delivered to an optional ``tap_callback(name, tensor, epocher)`` instead of a global writer; the feature flip before the
projector (:177-178) is skipped when the projector pools to (1,1) -- a global average is flip-invariant."""
from functools import partial
from typing import List, Union

import numpy as np
import torch
from torch import nn

from ...contrastyou.hooks.base import TrainerHook, EpocherHook
from ...contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss, SupConLoss1
from ...contrastyou.meters import AverageValueMeter
from ..arch.hook import SingleFeatureExtractor
from ..arch.unet import UNet
from ..epochers.helper import FixRandomSeed
from .utils import get_label, meter_focus

encoder_names = list(UNet.encoder_names)
decoder_names = list(UNet.decoder_names)


class PScheduler:
    def __init__(self, max_epoch, begin_value=0.0, end_value=1.0, p=0.5):
        self.max_epoch = max_epoch
        self.begin_value = float(begin_value)
        self.end_value = float(end_value)
        self.epoch = 0
        self.p = p

    def step(self):
        self.epoch += 1

    @property
    def value(self):
        return self.get_lr(self.epoch)

    def get_lr(self, cur_epoch):
        return self.begin_value + (self.end_value - self.begin_value) * np.power(cur_epoch / self.max_epoch, self.p)

    def state_dict(self):
        return {"epoch": self.epoch}

    def load_state_dict(self, sd):
        self.epoch = sd["epoch"]


class INFONCEHook(TrainerHook):
    @property
    def learnable_modules(self) -> List[nn.Module]:
        return [self._projector]

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, spatial_size=None,
                 data_name: str, contrast_on: str, sync_checks: bool = True, tap_callback=None) -> None:
        super().__init__(hook_name=name)
        assert feature_name in encoder_names + decoder_names, feature_name
        if feature_name not in encoder_names:
            raise NotImplementedError("dense (decoder) contrastive hooks are SURVEY row N3: not built")
        self._feature_name = feature_name
        self._weight = weight
        self._sync_checks = sync_checks
        self._tap_callback = tap_callback
        self._extractor = SingleFeatureExtractor(model, feature_name=feature_name)
        input_dim = model.get_channel_dim(feature_name)
        spatial_size = spatial_size or (1, 1)
        self._projector = self.init_projector(input_dim=input_dim, spatial_size=spatial_size)
        self._criterion = self.init_criterion()
        self._label_generator = partial(get_label, contrast_on=contrast_on, data_name=data_name)
        self._learnable_models = (self._projector,)

    def __call__(self):
        return _INFONCEEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                 projector=self._projector, criterion=self._criterion,
                                 label_generator=self._label_generator, tap_callback=self._tap_callback)

    def init_criterion(self) -> SupConLoss1:
        self._criterion = SupConLoss1(sync_checks=self._sync_checks)
        return self._criterion

    def init_projector(self, *, input_dim, spatial_size):
        return self.projector_class(input_dim=input_dim, hidden_dim=256, output_dim=256, head_type="mlp",
                                    normalize=True, spatial_size=spatial_size)

    @property
    def projector_class(self):
        from ...contrastyou.projectors.heads import ProjectionHead
        return ProjectionHead

    @property
    def is_encoder(self):
        return self._feature_name in encoder_names


class SelfPacedINFONCEHook(INFONCEHook):
    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, spatial_size=(1, 1),
                 data_name: str, contrast_on: str, mode="soft", p=0.5, begin_value=1e6, end_value=1e6,
                 correct_grad: bool = False, max_epoch: int, sync_checks: bool = True, tap_callback=None) -> None:
        self._mode = mode
        self._p = float(p)
        self._begin_value = float(begin_value)
        self._end_value = float(end_value)
        self._max_epoch = int(max_epoch)
        self._correct_grad = correct_grad
        super().__init__(name=name, model=model, feature_name=feature_name, weight=weight, spatial_size=spatial_size,
                         data_name=data_name, contrast_on=contrast_on, sync_checks=sync_checks,
                         tap_callback=tap_callback)

    def init_criterion(self) -> SelfPacedSupConLoss:
        self._scheduler = PScheduler(max_epoch=self._max_epoch, begin_value=self._begin_value,
                                     end_value=self._end_value, p=self._p)
        self._criterion = SelfPacedSupConLoss(weight_update=self._mode, correct_grad=self._correct_grad,
                                              sync_checks=self._sync_checks)
        return self._criterion

    def __call__(self):
        gamma = self._scheduler.value
        self._scheduler.step()
        self._criterion.set_gamma(gamma)
        return _SPINFONCEEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                   projector=self._projector, criterion=self._criterion,
                                   label_generator=self._label_generator, tap_callback=self._tap_callback)


class _INFONCEEpochHook(EpocherHook):
    def __init__(self, *, name: str, weight: float, extractor, projector,
                 criterion: Union[SupConLoss1, SelfPacedSupConLoss], label_generator, tap_callback=None) -> None:
        super().__init__(name)
        self._extractor = extractor
        self._extractor.bind()
        self._weight = weight
        self._projector = projector
        self._criterion = criterion
        self._label_generator = label_generator
        self._tap_callback = tap_callback
        self._n = 0
        self._label_cache = {}

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
        composition (the synthetic benchmark) costs no host->device copy."""
        key = (tuple(partition_group), tuple(label_group))
        t = self._label_cache.get(key)
        if t is None:
            labels = self._label_generator(partition_group=partition_group, label_group=label_group)
            t = torch.tensor(labels, dtype=torch.float32, device=device)
            if len(self._label_cache) < 64:
                self._label_cache[key] = t
        return t

    @meter_focus
    def __call__(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf, partition_group,
                 label_group, **kwargs):
        n_unl = len(unlabeled_logits_tf)
        feature_ = self._extractor.feature()
        if feature_.shape[0] != n_unl * 2:  # a slice costs a zero-fill + strided copy in backward: only when needed
            feature_ = feature_[-n_unl * 2:]
        unlabeled_features, unlabeled_tf_features = torch.chunk(feature_, 2, dim=0)
        pooled_global = tuple(getattr(self._projector, "_spatial_size", (1, 1))) == (1, 1)
        if not pooled_global:
            with FixRandomSeed(seed):
                unlabeled_features = torch.stack([affine_transformer(x) for x in unlabeled_features], dim=0)
            feature_ = torch.cat([unlabeled_features, unlabeled_tf_features], dim=0)
        norm_features_tf, norm_tf_features = torch.chunk(self._projector(feature_), 2)
        labels = self._labels(partition_group, label_group, feature_.device)
        loss = self._criterion(norm_features_tf, norm_tf_features, target=labels)
        self.meters["loss"].add(loss.detach())
        if self._n == 0 and self._tap_callback is not None:
            for tap in ("pos_mask", "sim_exp", "sim_logits"):
                self._tap_callback(tap, getattr(self._criterion, tap), self.epocher)
        self._n += 1
        return loss * self._weight

    def close(self):
        self._extractor.remove()


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
