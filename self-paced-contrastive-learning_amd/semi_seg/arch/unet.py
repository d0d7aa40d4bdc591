"""HIP-backed mirror of ``semi_seg/arch/unet.py``: same class surface (``UNet(input_dim, num_classes, max_channel,
momentum)``, ``forward(x, until)``, name tables, ``get_channel_dim``, ``set_grad``/``set_bn_track``), same submodule
names (``_Conv1.._Conv5, _Up5, _Up_conv5, ... _Deconv_1x1``) firing forward hooks, same ``state_dict`` keys
(``_ConvK.conv.{0,3}.weight``, ``_ConvK.conv.{1,4}.*``, ``_UpK.up.{1,2}.*``, ``_Deconv_1x1.*``).

The encoder blocks run as fused gfx950 kernels (functional.conv_block): NHWC storage, implicit-GEMM MFMA convolutions,
BatchNorm statistics in the conv epilogue, BN-apply + ReLU (+ 2x2 max-pool) fused into the neighbouring kernels.  The
tensors handed out (block outputs, ``until`` results, hook taps) are ordinary logical-NCHW tensors with channels-last
strides.  The decoder half (reference ``unet.py:193-230``, SURVEY row N1: fine-tune / evaluation path) runs from the
same kernels: ``_UpConv`` = nearest x2 upsample (a HIP streaming kernel; fusing it into the conv loader is listed in
DESIGN.md) + one fused conv-BN-ReLU (functional.conv_bn_relu), the skip concatenation is a HIP channel-interleave
kernel feeding the ordinary two-conv block, ``_Deconv_1x1`` is the HIP 1x1 convolution (functional.conv1x1) and
returns an fp32 class map.
"""
from collections import OrderedDict
from contextlib import contextmanager
from functools import lru_cache, partial
from typing import List

import torch
from torch import nn

from ... import config as _config
from ... import functional as F_hip

__all__ = ["UNet", "arch_order", "get_channel_dim", "sort_arch"]

_VIRTUAL_CAT = __import__("os").environ.get("SPCL_VIRTUAL_CAT", "1") != "0"  # A/B switch (see UNet._forward_blocks)
# narrowest skip that takes part.  A 16-channel pixel is a 32-byte run inside a 64-byte stride: the two producers then each
# write HALF of every line (bnrelu_fwd_lin 51 -> 88 us per step at 224^2) and the copy is cheaper; from 32 channels on the
# in-place halves win (fine-tune step, same box, three rounds: off 2.481 ms, >= 16: 2.455, >= 32: 2.439, >= 64: 2.449)
# Later (round 4): where the block's first convolution and its weight gradient can read the two tensors side by side and the
# gradient leaves as two dense tensors (functional.cat_pair_shape_ok: sizes the specialised kernels tile) that beats the halves
# of one buffer too -- 2.299 -> 2.261 ms with the 32-channel level moved over -- and is tried first (_CAT_PAIR_FIRST); the
# buffer remains for the other sizes.
_VIRTUAL_CAT_MINC = int(__import__("os").environ.get("SPCL_VIRTUAL_CAT_MINC", "32"))
_CAT_PAIR_FIRST = __import__("os").environ.get("SPCL_CAT_PAIR_FIRST", "1") != "0"  # A/B switch: 0 = the buffer wherever >= MINC
# widest level read as two tensors: the 64- and 128-channel levels work too (conv_fast slab by slab, the batched weight-gradient
# kernel block by block; tested) but measure the same as the halves of one buffer, +1..4 us: they keep the buffer
_CAT_PAIR_MAXC = int(__import__("os").environ.get("SPCL_CAT_PAIR_MAXC", "32"))
# the up-convolutions' BatchNorm + ReLU applied by the two-tensor convolution's and its weight gradient's loaders instead of
# a writer pass: built, bit-identical (tests/test_gpu_decoder.py) and 13 us SLOWER per step (2.336 -> 2.349 ms: the transform in
# two loaders costs more than the 27 us of writer passes it removes) -- off
_LAZY_UP = __import__("os").environ.get("SPCL_LAZY_UP", "0") != "0"
# the two-tensor level's dgrad also leaving the up-convolution's BatchNorm-backward sums (spcl_conv3x3_dgrad_split_bnstats):
# built, tested, and measured +4 us per step (2.3015 -> 2.3068 ms, three rounds): the epilogue costs what the reduction pass
# (20.8 + 11.9 us) cost -- off
_SPLIT_BNSTATS = __import__("os").environ.get("SPCL_SPLIT_BNSTATS", "0") != "0"
_LAZY_HEAD = __import__("os").environ.get("SPCL_LAZY_HEAD", "1") != "0"  # A/B switch: 0 writes the last activation and reads it back
_FUSED_UPSAMPLE = __import__("os").environ.get("SPCL_FUSED_UPSAMPLE", "1") != "0"  # A/B switch (BlockCfg.up2)
_ENCODER = ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5")
_DECODER = ("Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2", "Deconv_1x1")


def arch_order(name: str) -> int:
    return UNet.arch_elements.index(name)


def sort_arch(name_list: List[str], reverse=False) -> List[str]:
    return sorted(name_list, key=arch_order, reverse=reverse)


def _span(start, end, include_start, include_end):
    """names between start and end in architecture order (unet.py:34-64), with the reference's argument checks."""
    if start is None and include_start is False:
        raise ValueError("include_start should be True given start=None")
    if end is None and include_end is False:
        raise ValueError("include_end should be True given end=None")
    for v in (start, end):
        if isinstance(v, str) and v not in UNet.layer_dimension:
            raise ValueError(v)
    start, end = start or "Conv1", end or "Deconv_1x1"
    i0, i1 = arch_order(start), arch_order(end)
    if i0 > i1:
        raise ValueError((start, end))
    lo = i0 if include_start else i0 + 1
    hi = i1 + 1 if include_end else i1
    return [UNet.arch_elements[i] for i in range(lo, hi)]


class _ConvBlock(nn.Module):
    """Conv3x3(no bias) -> BatchNorm2d -> ReLU, twice (unet.py:67-82).  ``self.conv`` holds the parameters under the
    reference's Sequential indices; ``forward`` runs the fused HIP block, not the Sequential."""

    def __init__(self, in_ch, out_ch, momentum: float = 0.1, image_input: bool = False):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            nn.BatchNorm2d(out_ch, momentum=momentum),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_ch, out_ch, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            nn.BatchNorm2d(out_ch, momentum=momentum),
            nn.ReLU(inplace=True),
        )
        self._image_input = image_input
        self._compute_dtype = None  # None -> config default at call time
        self._plan = None           # (need_act, need_pool) set by UNet.forward for one call
        self._pooled = None
        self._link_in = self._link_out = None  # functional.PoolLink hand-over between consecutive encoder blocks
        self._act_dst = None        # where UNet.forward wants this call's activation written (half of a concat buffer)
        self._up2 = False           # this call's activation only feeds nn.Upsample(x2): return it upsampled (one launch less)
        self._x2_link = None        # functional.ActLink of the up-convolution whose RAW output this call's x2 is
        self._x2_bn = None          # functional.ActLink of the up-convolution whose ACTIVATION this call's x2 is
        self._up_link = None        # functional.UpLink shared with the up-convolution that reads this call's activation
        self._lazy = False          # this call's activation only feeds the 1x1 head, which applies BN + ReLU itself
        self._link_act = None       # functional.ActLink of a lazy call (UNet.forward hands it to the head)

    def _cfg(self, need_act, need_pool):
        bn_a, bn_b = self.conv[1], self.conv[4]
        dtype = self._compute_dtype or _config.get_compute_dtype()

        def use_batch_stats(bn):  # torch.nn.BatchNorm2d: batch statistics in train mode or without running stats
            return self.training or not bn.track_running_stats

        if use_batch_stats(bn_a) != use_batch_stats(bn_b):
            raise NotImplementedError("the two BatchNorm layers of a block must be in the same mode")
        track = tuple(self.training and bn.track_running_stats for bn in (bn_a, bn_b))
        bufs = tuple((bn.running_mean, bn.running_var, bn.num_batches_tracked) for bn in (bn_a, bn_b))
        assert bn_a.momentum is not None and bn_a.eps == bn_b.eps
        return F_hip.BlockCfg(dtype, use_batch_stats(bn_a), float(bn_a.momentum), float(bn_a.eps), track, need_act,
                              need_pool, self._image_input, bufs)

    def forward(self, x, x2=None):
        """``x2``: the input is ``torch.cat((x, x2), 1)``, read from the two tensors in place (functional.cat_pair_supported)"""
        need_act, need_pool = self._plan if self._plan is not None else (True, False)
        self._plan = None
        if len(self._forward_hooks) > 0:
            need_act = True  # a forward hook (arch/hook.py feature tap) wants the block output
        c = self.conv
        cfg = self._cfg(need_act, need_pool)
        cfg.link_in, self._link_in = self._link_in, None
        cfg.act_dst, self._act_dst = self._act_dst, None
        cfg.up2, self._up2 = (self._up2 and len(self._forward_hooks) == 0), False
        cfg.lazy_act, self._lazy = (self._lazy and len(self._forward_hooks) == 0), False
        cfg.up_link, self._up_link = self._up_link, None
        cfg.x2_link, self._x2_link = (self._x2_link if x2 is not None else None), None
        cfg.x2_bn, self._x2_bn = (self._x2_bn if x2 is not None else None), None
        cfg.want_gap = len(self._forward_hooks) > 0  # a feature tap: its projector pools globally (functional.conv_block)
        act, pooled = F_hip.conv_block(x, c[0].weight, c[1].weight, c[1].bias, c[3].weight, c[4].weight, c[4].bias, cfg, x2)
        self._pooled = pooled
        self._link_out = cfg.link_out  # for the block that consumes the pooled output (UNet.forward hands it over)
        self._link_act = cfg.link_act
        return act

    def take_pooled(self):
        p, self._pooled = self._pooled, None
        return p


class _UpConv(nn.Module):
    """Upsample(x2, nearest) -> Conv3x3 -> BN -> ReLU (unet.py:85-97)."""

    def __init__(self, in_ch, out_ch, momentum=0.1):
        super().__init__()
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2),
            nn.Conv2d(in_ch, out_ch, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False),
            nn.BatchNorm2d(out_ch, momentum=momentum),
            nn.ReLU(inplace=True),
        )

        self._compute_dtype = None
        self._act_dst = None
        self._link_act = None
        self._bn_link = None  # functional.ActLink of the last call: (raw output, coefficients), for the consumer's dgrad

    def forward(self, x, pre_upsampled: bool = False, lazy: bool = False, virtual_up: bool = False, up_link=None):
        """``pre_upsampled``: ``x`` already is the x2-upsampled tensor (the producing block wrote it that way, BlockCfg.up2);
        ``lazy``: return the RAW convolution output -- the only consumer applies this BatchNorm + ReLU itself and takes the
        coefficients from ``self._link_act`` (UNet.forward hands it to the next block: functional.ActLink)"""
        bn = self.up[2]
        dtype = self._compute_dtype or _config.get_compute_dtype()
        training = self.training or not bn.track_running_stats
        cfg = F_hip.BlockCfg(dtype, training, float(bn.momentum), float(bn.eps),
                             (self.training and bn.track_running_stats,), True, False, False,
                             ((bn.running_mean, bn.running_var, bn.num_batches_tracked),))
        cfg.act_dst, self._act_dst = self._act_dst, None
        cfg.lazy_act = bool(lazy) and cfg.act_dst is None
        cfg.up_in = bool(virtual_up)  # the convolution and its weight gradient read nn.Upsample(x2)(x) from x itself
        cfg.up_link = up_link if virtual_up else None
        if not pre_upsampled and not virtual_up:
            x = F_hip.upsample2x(x, dtype)  # nn.Upsample(scale_factor=2), nearest
        out = F_hip.conv_bn_relu(x, self.up[1].weight, bn.weight, bn.bias, cfg)
        self._link_act = cfg.link_act
        self._bn_link = cfg.bn_link
        return out


class _Conv1x1(nn.Conv2d):
    """``_Deconv_1x1`` (unet.py:147): an nn.Conv2d(prev, num_classes, 1) whose forward is the HIP head kernel; the fp32
    class map comes back as a logical [N,K,H,W] view over [N,H,W,K] storage."""
    _compute_dtype = None

    _link = None  # set by UNet.forward for one call: the producing block returned its RAW output (functional.ActLink)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("self-paced-contrastive-learning_amd runs on MI355X only: got a CPU tensor "
                               "(the HIP path has no CPU fallback)")
        link, self._link = self._link, None
        if link is not None:
            return F_hip.conv1x1_bn(x, self.weight, self.bias, link)
        return F_hip.conv1x1(x, self.weight, self.bias, self._compute_dtype or _config.get_compute_dtype())


class UNet(nn.Module):
    layer_dimension = {"Conv1": 1, "Conv2": 2, "Conv3": 4, "Conv4": 8, "Conv5": 16, "Up_conv5": 8, "Up_conv4": 4,
                       "Up_conv3": 2, "Up_conv2": 1, "Deconv_1x1": None}
    encoder_names = _ENCODER
    decoder_names = _DECODER
    arch_elements = tuple(list(_ENCODER) + list(_DECODER))

    def __init__(self, input_dim=3, num_classes=1, max_channel=256, momentum=0.1):
        super().__init__()
        self._input_dim = input_dim
        self._num_classes = num_classes
        assert max_channel % 16 == 0 and max_channel >= 128, max_channel
        self._max_channel = max_channel
        for i in range(1, 5):
            setattr(self, f"_max_pool{i}", nn.MaxPool2d(kernel_size=2, stride=2))  # fused into the blocks
        ch = self.get_channel_dim
        prev = input_dim
        for k, name in enumerate(_ENCODER):
            setattr(self, "_" + name, _ConvBlock(prev, ch(name), momentum=momentum, image_input=(k == 0)))
            prev = ch(name)
        for lvl in (5, 4, 3, 2):
            co = ch(f"Up_conv{lvl}")
            setattr(self, f"_Up{lvl}", _UpConv(prev, co, momentum=momentum))
            setattr(self, f"_Up_conv{lvl}", _ConvBlock(prev, co, momentum=momentum))
            prev = co
        self._Deconv_1x1 = _Conv1x1(prev, num_classes, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0))
        self._boundary_hooks = {}  # encoder block name -> callable run in backward at that block's output (see forward)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, until: str = None):
        if until:
            if until not in self.layer_dimension:
                raise KeyError(f"`return_until` should be in {', '.join(self.layer_dimension.keys())},"
                               f" given {until}  ")
        encoder_only = until in _ENCODER
        if x.is_cuda:
            F_hip.bn_acc_arena_begin(x.device)  # one zero fill for every BatchNorm accumulator block of this pass (and its backward)
        self._prepack(x, until)
        self._prepare_eval_bn(x, until)
        try:
            return self._forward_blocks(x, until, encoder_only)
        finally:
            # whatever this pass announced and did not consume (an exception mid-forward, a size the announcement got
            # wrong) must not outlive it: the packed copies are of THIS step's weights
            F_hip.clear_prepacked()

    def _forward_blocks(self, x, until, encoder_only):
        e = x
        skips = {}
        # torch.cat((skip, up), dim=1) of the decoder (unet.py:194-224) without a copy: both producers -- the encoder block's
        # BN+ReLU writer and the up-convolution's -- write their half of ONE [N, H, W, 2C] buffer, the concatenation is a
        # view of it and its gradient is read back half by half in place (functional.virtual_cat; SPCL_VIRTUAL_CAT=0: the
        # copying kernels).  Not for a block with a forward hook (a feature tap wants a dense tensor of its own).
        cats = {}
        pairs = set()  # skips whose level reads the two tensors side by side (decided here: the up-convolution may then be lazy)
        if not encoder_only and x.is_cuda and x.dim() == 4:
            H0, W0 = int(x.shape[2]), int(x.shape[3])
            for k, name in enumerate(_ENCODER[:-1]):
                blk = getattr(self, "_" + name)
                c = self.get_channel_dim(name)
                h, w = H0 >> k, W0 >> k
                if not (c % 16 == 0 and len(blk._forward_hooks) == 0 and h >= 2 and w >= 2 and h % 2 == 0 and w % 2 == 0):
                    continue
                dt = blk._compute_dtype or _config.get_compute_dtype()
                ub = getattr(self, f"_Up_conv{k + 2}")
                if (_CAT_PAIR_FIRST and c <= _CAT_PAIR_MAXC and len(ub._forward_pre_hooks) == 0
                        and F_hip.cat_pair_shape_ok(int(x.shape[0]), c, h, w, c, ub._compute_dtype or dt)):
                    pairs.add(name)  # this level's block reads the two tensors side by side (below): no buffer
                elif _VIRTUAL_CAT and c >= _VIRTUAL_CAT_MINC:
                    cats[name] = F_hip.cat_buffer(int(x.shape[0]), h, w, c, c, dt, x.device)
        for k, name in enumerate(_ENCODER):
            blk = getattr(self, "_" + name)
            is_last = (until == name) or k == len(_ENCODER) - 1
            blk._plan = (is_last or not encoder_only, not is_last)  # the decoder needs every block output (skips)
            if name in cats and until != name:
                blk._act_dst = cats[name][1]
            up2_first = (not encoder_only and k == len(_ENCODER) - 1 and x.is_cuda and _FUSED_UPSAMPLE
                         and len(blk._forward_hooks) == 0 and len(self._Up5._forward_hooks) == 0
                         and len(self._Up5.up[0]._forward_hooks) == 0)
            virt_first = False
            if up2_first:  # ... or not written upsampled at all: Up5's loaders read it at half resolution (functional.up_in_shape_ok)
                hh, ww = int(x.shape[2]) >> k, int(x.shape[3]) >> k
                virt_first = F_hip.up_in_shape_ok(int(x.shape[0]), self.get_channel_dim(name), 2 * hh, 2 * ww,
                                                  self.get_channel_dim("Up_conv5"),
                                                  self._Up5._compute_dtype or _config.get_compute_dtype())
            blk._up2 = up2_first and not virt_first  # Conv5's activation only feeds Up5's nn.Upsample: written x2-upsampled directly
            up_link = F_hip.UpLink() if (up2_first and virt_first) else None
            blk._up_link = up_link
            if k > 0:
                prev = getattr(self, "_" + _ENCODER[k - 1])
                blk._link_in, prev._link_out = prev._link_out, None
            out = blk(e)
            if until == name:
                return out
            skips[name] = out
            e = blk.take_pooled()
            cb = self._boundary_hooks.get(name)
            if cb is not None and e is not None and e.requires_grad:
                # fires when backward has produced the gradient of this block's pooled output, i.e. once every later
                # block has been differentiated (ddp.enable_unet_overlap starts the early gradient bucket there)
                e.register_hook(lambda g, cb=cb: cb() and None)
        # decoding + concat path (unet.py:193-230)
        d = skips["Conv5"]
        pre_up, virt_up = up2_first and not virt_first, up2_first and virt_first
        for lvl, skip in ((5, "Conv4"), (4, "Conv3"), (3, "Conv2"), (2, "Conv1")):
            up = getattr(self, f"_Up{lvl}")
            if skip in cats:
                up._act_dst = cats[skip][2]
            blk = getattr(self, f"_Up_conv{lvl}")
            # the up-convolution's activation goes nowhere but into this level's two-tensor convolution: never written, its
            # BatchNorm + ReLU applied by that convolution's (and its weight gradient's) loader
            lazy_up = (_LAZY_UP and skip in pairs and until != f"Up{lvl}" and len(up._forward_hooks) == 0
                       and all(len(m._forward_hooks) == 0 for m in up.up) and len(blk._forward_pre_hooks) == 0
                       and self.get_channel_dim(skip) <= 32
                       and (up._compute_dtype or _config.get_compute_dtype()) ==
                           (blk._compute_dtype or _config.get_compute_dtype()))
            d = up(d, pre_upsampled=pre_up, lazy=lazy_up, virtual_up=virt_up, up_link=up_link)
            lazy_up = lazy_up and up._link_act is not None
            # this decoder block's activation feeds the next level's nn.Upsample only (not the last block, not the `until` one)
            nxt = getattr(self, f"_Up{lvl - 1}", None) if lvl > 2 else None
            pre_up = (nxt is not None and until != f"Up_conv{lvl}" and _FUSED_UPSAMPLE and d.is_cuda
                      and len(blk._forward_hooks) == 0 and len(nxt._forward_hooks) == 0
                      and len(nxt.up[0]._forward_hooks) == 0)
            virt_up = False
            if pre_up:  # ... read at half resolution by the next up-convolution's loaders instead, where they can
                virt_up = F_hip.up_in_shape_ok(int(d.shape[0]), self.get_channel_dim(f"Up_conv{lvl}"), 2 * int(d.shape[2]),
                                               2 * int(d.shape[3]), self.get_channel_dim(f"Up_conv{lvl - 1}"),
                                               nxt._compute_dtype or _config.get_compute_dtype())
                pre_up = not virt_up
            blk._up2 = pre_up
            up_link = F_hip.UpLink() if virt_up else None
            blk._up_link = up_link
            # the last block's activation feeds the 1x1 head only: the head applies that BatchNorm + ReLU in its own loader
            # (forward and backward) and leaves the BatchNorm-backward sums; no activation tensor, two passes less
            blk._lazy = (lvl == 2 and until is None and _LAZY_HEAD and d.is_cuda and len(blk._forward_hooks) == 0
                         and len(self._Deconv_1x1._forward_hooks) == 0 and len(self._Deconv_1x1._forward_pre_hooks) == 0
                         and blk.conv[3].weight.shape[0] % 16 == 0
                         and (blk._compute_dtype or _config.get_compute_dtype()) ==
                             (self._Deconv_1x1._compute_dtype or _config.get_compute_dtype()))
            bdt = blk._compute_dtype or _config.get_compute_dtype()
            if skip in cats:
                d = blk(F_hip.virtual_cat(skips[skip], d, cats[skip][0]))
            elif lazy_up:
                assert F_hip.cat_pair_supported(skips[skip], d, blk.conv[0].weight.shape[0], bdt), "planned two-tensor level"
                blk._x2_link, up._link_act = up._link_act, None
                d = blk(skips[skip], x2=d)
            elif (len(blk._forward_pre_hooks) == 0 and torch.is_tensor(skips[skip])
                  and F_hip.cat_pair_supported(skips[skip], d, blk.conv[0].weight.shape[0], bdt)):
                # the narrow level (16 + 16 channels: too narrow for the producers to write halves of one buffer -- half a
                # cache line per pixel): both tensors stay where they are, the block's first convolution and its weight
                # gradient read them side by side, the gradient comes back as one tensor read half by half
                if _SPLIT_BNSTATS and not any(len(m._forward_hooks) for m in (up, *up.up)):
                    blk._x2_bn = up._bn_link  # (its dgrad then leaves the up-convolution's BatchNorm-backward sums too)
                d = blk(skips[skip], x2=d)
            else:
                d = blk(F_hip.concat_channels(skips[skip], d, bdt))
            if until == f"Up_conv{lvl}":
                return d
        self._Deconv_1x1._link, blk._link_act = blk._link_act, None
        return self._Deconv_1x1(d)

    def _prepare_eval_bn(self, x, until):
        """an eval-mode pass (validation; a frozen block under ``disable_bn``): the coefficients of every BatchNorm it applies,
        from the running statistics, in ONE launch up front (functional.prepare_eval_affines) instead of one per layer"""
        if not x.is_cuda or x.dim() != 4:
            return
        bns = []
        for name in _ENCODER:
            m = getattr(self, "_" + name)
            if not m.training:
                bns.extend([m.conv[1], m.conv[4]])
            if until == name:
                break
        else:
            for lvl in (5, 4, 3, 2):
                up, blk = getattr(self, f"_Up{lvl}"), getattr(self, f"_Up_conv{lvl}")
                if not up.training:
                    bns.append(up.up[2])
                if until == f"Up{lvl}":
                    break
                if not blk.training:
                    bns.extend([blk.conv[1], blk.conv[4]])
                if until == f"Up_conv{lvl}":
                    break
        bns = [b for b in bns if isinstance(b, nn.BatchNorm2d)]
        if len(bns) > 1:
            F_hip.prepare_eval_affines(bns)

    def _prepack(self, x, until):
        """announce the 3x3 convolutions this forward pass runs (and the image size each runs at): their weights are packed
        into MFMA fragment order by ONE launch up front (functional.prepack_weights) instead of one launch per block"""
        if not x.is_cuda or x.dim() != 4:
            return
        H, W = int(x.shape[2]), int(x.shape[3])
        layers, dtypes = [], set()

        def block(m, h, w):
            dtypes.add(m._compute_dtype or _config.get_compute_dtype())
            layers.extend([(m.conv[0].weight, h, w), (m.conv[3].weight, h, w)])

        done = False
        for k, name in enumerate(_ENCODER):
            block(getattr(self, "_" + name), H >> k, W >> k)
            if until == name:
                done = True
                break
        if not done:
            h, w = H >> (len(_ENCODER) - 1), W >> (len(_ENCODER) - 1)
            for lvl in (5, 4, 3, 2):
                h, w = 2 * h, 2 * w
                up = getattr(self, f"_Up{lvl}")
                dtypes.add(up._compute_dtype or _config.get_compute_dtype())
                layers.append((up.up[1].weight, h, w))
                block(getattr(self, f"_Up_conv{lvl}"), h, w)
                if until == f"Up_conv{lvl}":
                    break
        if len(dtypes) == 1:
            dtype = dtypes.pop()
            # the first block's "image3" backward wants the autocorrelation of the one-channel input: same launch
            c1 = self._Conv1.conv
            image = None
            if (x.shape[1] == 1 and x.dtype == torch.float32 and x.is_contiguous() and torch.is_grad_enabled()
                    and c1[0].weight.requires_grad and not x.requires_grad and self._Conv1.training
                    and F_hip._image3_supported(self._Conv1._cfg(True, False), 1, F_hip._n.dtype_code(dtype), int(x.shape[0]),
                                                H, W, F_hip._ru16(c1[0].weight.shape[0]))):
                image = x.detach().view(x.shape[0], H, W)
            F_hip.prepack_weights(layers, dtype, image)

    def set_compute_dtype(self, dtype):
        """torch.float32 (parity mode) or torch.bfloat16 for all fused blocks of this network."""
        for m in self.modules():
            if isinstance(m, (_ConvBlock, _UpConv, _Conv1x1)):
                m._compute_dtype = dtype
        return self

    # ------------------------------------------------------------------------------------------ reference API
    @lru_cache()
    def get_channel_dim(self, name: str):
        if name == "Deconv_1x1":
            return self._num_classes
        elif name in self.layer_dimension:
            return int(self.layer_dimension[name] / 16 * self._max_channel)
        else:
            raise KeyError(name)

    @contextmanager
    def set_grad(self, enable=True, *, start: str = None, end: str = None, include_start=True, include_end=True):
        comps = _span(start, end, include_start, include_end)
        prev = OrderedDict()
        for c in comps:
            m = getattr(self, "_" + c)
            states = {p.requires_grad for p in m.parameters()}
            if len(states) != 1:
                raise RuntimeError(f"mixed requires_grad state in {c}")
            prev[c] = states.pop()
            m.requires_grad_(enable)
        yield self
        for c in comps:
            getattr(self, "_" + c).requires_grad_(prev[c])

    @contextmanager
    def set_bn_track(self, enable=True, *, start: str = None, end: str = None, include_start=True, include_end=True):
        comps = _span(start, end, include_start, include_end)

        def switch(m, enable=True):
            if hasattr(m, "track_running_stats"):
                m.track_running_stats = enable

        prev = OrderedDict()
        for c in comps:
            m = getattr(self, "_" + c)
            states = {s.track_running_stats for s in m.modules() if hasattr(s, "track_running_stats")}
            if len(states) != 1:
                continue
            prev[c] = states.pop()
            m.apply(partial(switch, enable=enable))
        yield self
        for c, st in prev.items():
            getattr(self, "_" + c).apply(partial(switch, enable=st))

    @property
    def num_classes(self):
        return self._num_classes


def get_channel_dim(layer_name: str, *, max_channel=None):
    max_channel = max_channel or 256
    assert layer_name in {k: v for k, v in UNet.layer_dimension.items() if v is not None}
    return int(UNet.layer_dimension[layer_name] / 16 * max_channel)
