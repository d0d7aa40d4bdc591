"""Adaptor for the reference's OLD, ``init()``-driven pre-train API so that
``semi_seg.main_infonce``-style drivers reach the same HIP kernels:

    epocher = InfoNCEPretrainEpocher(model=..., optimizer=..., chain_dataloader=..., feature_names=[...], ...)
    epocher.init(reg_weight=w, projectors_wrapper=wrapper, infoNCE_criterion=[crit, ...])   # comparable.py:257-268
    epocher.set_global_contrast_method(contrast_on_list=[...])                               # :270-279
    epocher.run()

Restated from ``_InfoNCEBasedEpocher`` / ``InfoNCEEpocher`` (semi_seg/epochers/comparable.py:250-450), the old
``_PretrainEpocherMixin`` (semi_seg/epochers/_mixins.py:181-274: hard-coded ``until="Conv5"``, NO flip of the second
view's images, ``unlabeled_logits_tf=unlabel_tf_logits``) and ``ContrastiveProjectorWrapper`` (semi_seg/utils.py:55-117).
Their deepclustering2 base classes (``_Epocher``, ``Trainer``) are out of scope; ``get_config`` look-ups are replaced by
constructor arguments (``feature_names``, ``data_name``, ``max_channel``)."""
from itertools import cycle
from typing import List, Sequence, Union

import torch
from torch import nn

from ...contrastyou import meters as _meters
from ...contrastyou.losses.contrast_loss3 import SupConLoss1
from ...contrastyou.meters import AverageValueMeter
from ...contrastyou.projectors.heads import DenseProjectionHead, ProjectionHead
from ..arch.hook import FeatureExtractor
from ..arch.unet import get_channel_dim
from .helper import (ACDCCycleGenerator, FixRandomSeed, PartitionLabelGenerator, PatientLabelGenerator, SIMCLRGenerator)
from .pretrain import PretrainEncoderEpocher, unzip_twice_transformed


def _nlist(n):
    return lambda v: list(v) if isinstance(v, (list, tuple)) and not (len(v) == 2 and all(isinstance(e, int) for e in v)
                                                                     and n != 2) else [v] * n


class ContrastiveProjectorWrapper(nn.Module):
    """semi_seg/utils.py:55-117: an ordered collection of projection heads keyed ``"<index>|<feature name>"``; iterating
    yields the heads in registration order (global ones first when registered first, as InfoNCETrainer does)."""

    def __init__(self, max_channel: int = 256):
        super().__init__()
        self._projectors = nn.ModuleDict()
        self._max_channel = max_channel
        self._index = 0
        self._global_feature_names: List[str] = []
        self._dense_feature_names: List[str] = []

    def _register(self, feature_name, projector):
        self._projectors[f"{self._index}|{feature_name}"] = projector
        self._index += 1

    def register_global_projector(self, *, feature_names: Union[str, List[str]], head_type="mlp", output_dim=256,
                                  normalize=True, pool_name="adaptive_avg", **kwargs):
        names = [feature_names] if isinstance(feature_names, str) else list(feature_names)
        self._global_feature_names = names
        pair = _nlist(len(names))
        for f, h, nm, p, o in zip(names, pair(head_type), pair(normalize), pair(pool_name), pair(output_dim)):
            self._register(f, ProjectionHead(input_dim=get_channel_dim(f, max_channel=self._max_channel), head_type=h,
                                             normalize=nm, pool_name=p, output_dim=o))

    def register_dense_projector(self, *, feature_names, output_dim=64, head_type, normalize=False,
                                 pool_name="adaptive_avg", spatial_size=(16, 16), **kwargs):
        names = [feature_names] if isinstance(feature_names, str) else list(feature_names)
        self._dense_feature_names = names
        pair = _nlist(len(names))
        sizes = [spatial_size] * len(names) if isinstance(spatial_size[0], int) else list(spatial_size)
        for f, h, nm, p, o, sz in zip(names, pair(head_type), pair(normalize), pair(pool_name), pair(output_dim), sizes):
            self._register(f, DenseProjectionHead(input_dim=get_channel_dim(f, max_channel=self._max_channel),
                                                  output_dim=o, head_type=h, normalize=nm, pool_name=p, spatial_size=sz))

    @property
    def feature_names(self):
        return [k.split("|", 1)[1] for k in self._projectors.keys()]

    def __len__(self):
        return len(self._projectors)

    def __iter__(self):
        return iter(self._projectors.values())

    def __getitem__(self, i):
        return list(self._projectors.values())[i]


def _label_generator(data_name: str, contrast_on: str):
    """comparable.py:298-337"""
    table = {"partition": PartitionLabelGenerator, "patient": PatientLabelGenerator, "self": SIMCLRGenerator}
    if data_name == "acdc":
        table = dict(table, cycle=ACDCCycleGenerator)
    if contrast_on not in table:
        raise NotImplementedError(contrast_on)
    return table[contrast_on]()


class InfoNCEPretrainEpocher(PretrainEncoderEpocher):
    """old-API pre-train epocher: the projectors and criteria arrive through ``init()``, one (projector, criterion,
    contrast_on) triple per feature position, the regularisation is their importance-weighted average."""

    def __init__(self, *, feature_names: Union[str, Sequence[str]], feature_importance=None, data_name="acdc",
                 dense_pool_method="adaptive_avg", **kwargs):
        names = [feature_names] if isinstance(feature_names, str) else list(feature_names)
        # _mixins.py:262 runs the encoder only; a decoder feature position (test/test_infonce.py:21: "Up_conv2") needs the
        # network run that far: until = the deepest requested position
        from ..arch.unet import UNet
        order = list(UNet.arch_elements)
        kwargs.setdefault("inference_until", max(names + ["Conv5"], key=order.index))
        self._dense_pool_method = dense_pool_method  # config["ProjectorParams"]["DenseParams"]["pool_method"] (:501)
        super().__init__(**kwargs)
        self._feature_position = [feature_names] if isinstance(feature_names, str) else list(feature_names)
        self._feature_importance = list(feature_importance) if feature_importance is not None else \
            [1.0] * len(self._feature_position)
        assert len(self._feature_importance) == len(self._feature_position)
        self._data_name = data_name
        self._fextractor = FeatureExtractor(self._model, self._feature_position)
        self._initialized = self._contrast_set = False
        with self.meters.focus_on(self.meter_focus):
            self.meters.register_meter("mi", AverageValueMeter())
            for i, p in enumerate(self._feature_position):
                self.meters.register_meter(f"mi_{p}|{i}", AverageValueMeter())  # "individual_mis" (comparable.py:284)

    # ---- comparable.py:257-268
    def init(self, *, reg_weight: float, projectors_wrapper: ContrastiveProjectorWrapper = None,
             infoNCE_criterion: List[nn.Module] = None, **kwargs):
        assert projectors_wrapper is not None and infoNCE_criterion is not None, (projectors_wrapper, infoNCE_criterion)
        n_global = len([p for p in projectors_wrapper if isinstance(p, ProjectionHead)])
        assert n_global == len(infoNCE_criterion), (n_global, len(infoNCE_criterion))
        assert len(projectors_wrapper) == len(self._feature_position), (len(projectors_wrapper), self._feature_position)
        self._reg_weight = float(reg_weight)
        self._projectors_wrapper = projectors_wrapper
        self._encoder_criterion_generator = cycle(infoNCE_criterion)
        self._normal_criterion = SupConLoss1()
        self._initialized = True

    # ---- comparable.py:270-279
    def set_global_contrast_method(self, *, contrast_on_list):
        assert isinstance(contrast_on_list, (tuple, list))
        for e in contrast_on_list:
            assert e in ("partition", "patient", "cycle", "self"), e
        self._encoder_contrastive_name_generator = cycle(list(contrast_on_list))
        self._contrast_set = True

    def run(self):  # :287-290
        if not self._contrast_set:
            raise RuntimeError(f"`set_global_contrast_method` should be called first for {self.__class__.__name__}.")
        if not self._initialized:
            raise RuntimeError("`init(reg_weight=, projectors_wrapper=, infoNCE_criterion=)` should be called first")
        self._fextractor.bind()
        try:
            return super().run()
        finally:
            self._fextractor.remove()

    # ---- _mixins.py:225-260: no image flip; the hook wire format is not used on this path
    def step_compute(self, data, seed=None):
        import random
        seed = random.randint(0, int(1e7)) if seed is None else seed
        _meters.begin_batch()
        (image, image_tf), _, filename, partitions, groups = unzip_twice_transformed(data, self._device)
        self._fextractor.clear()
        self._fextractor.set_enable(True)
        self._model(torch.cat([image, image_tf], dim=0), until=self._inference_until)
        self._fextractor.set_enable(False)
        reg_loss = self._regularization(n_unl=len(image), seed=seed, label_group=groups, partition_group=partitions)
        total_loss = reg_loss * self._reg_weight if self._reg_weight != 1.0 else reg_loss
        if self._flat_params is not None:
            self._flat_params.zero_grad()
            total_loss.backward(gradient=self._unit_grad(total_loss))
            self._flat_params.gather_grads()
        else:
            self._optimizer.zero_grad(set_to_none=True)
            total_loss.backward(gradient=self._unit_grad(total_loss))
        return reg_loss

    # ---- comparable.py:347-364
    def _regularization(self, *, n_unl, seed, label_group, partition_group, **kwargs):
        losses = []
        for name, feature, projector in zip(self._feature_position, self._fextractor, self._projectors_wrapper):
            feature = feature if feature.shape[0] == 2 * n_unl else feature[-2 * n_unl:]
            losses.append(self.generate_infonce(feature_name=name, features=feature, projector=projector,
                                                seed=seed, partition_group=partition_group, label_group=label_group))
        total = None
        for l, w in zip(losses, self._feature_importance):  # weighted_average_iter (contrastyou/utils/utils.py:66-68)
            total = l * w if total is None else total + l * w
        reg_loss = total / (sum(self._feature_importance) + 1e-16)
        self.meters["mi"].add(-reg_loss.detach())
        for i, (p, l) in enumerate(zip(self._feature_position, losses)):
            self.meters[f"mi_{p}|{i}"].add(-l.detach())
        return reg_loss

    # ---- comparable.py:292-304
    def unlabeled_projection(self, unl_features, projector, seed):
        first, second = torch.chunk(unl_features, 2, dim=0)
        if not (isinstance(projector, ProjectionHead) and tuple(projector._spatial_size) == (1, 1)):
            with FixRandomSeed(seed):  # (a global pool is flip-invariant: only the dense heads need the flipped copy)
                first = self._affine_transformer.apply_batch(first)
        proj_tf_feature, proj_feature_tf = torch.chunk(projector(torch.cat([second.contiguous(), first], dim=0)), 2, dim=0)
        return proj_tf_feature, proj_feature_tf

    # ---- comparable.py:366-387, 415-450
    def generate_infonce(self, *, feature_name, features, projector, seed, partition_group, label_group):
        proj_tf_feature, proj_feature_tf = self.unlabeled_projection(features, projector, seed)
        if not isinstance(projector, ProjectionHead):  # "it goes to a **dense** representation on pixels" (:380-387)
            return self._dense_based_infonce(feature_name=feature_name, proj_tf_feature=proj_tf_feature,
                                             proj_feature_tf=proj_feature_tf, projector=projector)
        assert proj_tf_feature.dim() == 2, proj_tf_feature.shape
        contrast_on = next(self._encoder_contrastive_name_generator)
        criterion = next(self._encoder_criterion_generator)
        gen = _label_generator("acdc" if self._data_name == "acdc" else "prostate", contrast_on)
        if self._data_name == "acdc":
            labels = gen(partition_list=partition_group, patient_list=[p.split("_")[0] for p in label_group],
                         experiment_list=[p.split("_")[1] for p in label_group])
        elif self._data_name == "prostate":
            labels = gen(partition_list=partition_group, patient_list=[p.split("_")[0] for p in label_group])
        else:
            labels = gen(partition_list=partition_group, patient_list=label_group)
        return criterion(proj_feature_tf, proj_tf_feature, target=labels)

    # ---- comparable.py:452-533: the dense branch.  Every pixel of the (pooled) dense projection is its own class, its
    # positive the same pixel of the other view (``SupConLoss1`` without target: SimCLR identity positives).  The reference
    # asks ``is_normalized`` (a device -> host readback) whether to normalise; here that is known from how the map was made:
    # a head that normalises hands over unit pixels, a pooling that changes the size breaks them.
    dense_output_size = (12, 12)  # comparable.py:500

    def _dense_based_infonce(self, *, feature_name, proj_tf_feature, proj_feature_tf, projector):
        from ... import functional as F_hip
        unit = bool(getattr(projector, "_normalize", False))
        if "Conv" in feature_name:  # _dense_infonce_for_encoder (:471-492): no spatial neighbourhood, pixels as they come
            def rows(x):
                x = x if unit else F_hip.l2norm_channels(x)  # Normalize(dim=2) of the [b, hw, c] view
                return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])  # _reshape_dense_feature + reshape(-1, c)
            return self._normal_criterion(rows(proj_feature_tf), rows(proj_tf_feature))
        assert "Up" in feature_name, feature_name  # _dense_infonce_for_decoder (:494-513)
        if self._dense_pool_method not in ("adaptive_avg", "adaptive_max"):
            raise NotImplementedError(f"dense pool_method {self._dense_pool_method!r}: adaptive_avg / adaptive_max are on the "
                                      "HIP path (the reference's third choice, bilinear resizing, is not)")

        def tailored(x):  # _dense_featuremap_tailoring (:515-532): resize to 12 x 12, unit pixels
            if tuple(x.shape[2:]) != tuple(self.dense_output_size):
                x = F_hip.adaptive_pool2d(x, self.dense_output_size, "max" if self._dense_pool_method == "adaptive_max" else "avg")
                return F_hip.l2norm_channels(x)
            return x if unit else F_hip.l2norm_channels(x)

        def rows(x):
            return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        return self._normal_criterion(rows(tailored(proj_tf_feature)), rows(tailored(proj_feature_tf)))
