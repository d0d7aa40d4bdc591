"""CPU-only tests of the host-side mirror: schedule, label generators, flips, meters, hook plumbing, state_dict
keys, trainer schedule.  (Compute classes raise on CPU tensors by design; only their host logic is exercised.)"""
import math
import random

import numpy as np
import pytest
import torch

import spcl_amd  # noqa: F401
from oracle import spcl_oracle as O
from spcl_amd.contrastyou.hooks.base import CombineEpochHook, EpocherHook, TrainerHook
from spcl_amd.contrastyou.meters import AverageValueMeter, MeterInterface
from spcl_amd.semi_seg.arch import SingleFeatureExtractor, UNet
from spcl_amd.semi_seg.epochers.helper import FixRandomSeed, TensorRandomFlip
from spcl_amd.semi_seg.hooks import PScheduler, create_infonce_hooks, create_sp_infonce_hooks, feature_until_from_hooks
from spcl_amd.semi_seg.hooks.utils import get_label
from spcl_amd.semi_seg.trainers import WarmupCosine


def test_pscheduler_matches_oracle_and_closed_form():
    s, o = PScheduler(80, 3, 70, 0.5), O.PScheduler(80, 3, 70, 0.5)
    for e in range(80):
        assert s.value == pytest.approx(o.value, rel=1e-12)
        assert s.value == pytest.approx(3 + 67 * math.sqrt(e / 80), rel=1e-12)
        s.step()
        o.step()


def test_label_generators_match_oracle():
    groups = ["patient004_00", "patient004_01", "patient001_00", "patient010_01", "patient001_01"]
    parts = ["2", "0", "1", "0", "2"]
    for on in ("partition", "patient", "cycle", "self"):
        assert get_label(on, "acdc", parts, groups) == O.get_label(on, "acdc", parts, groups)
    for on in ("partition", "patient", "self"):
        assert get_label(on, "prostate", parts, ["Case00_1", "Case03_0", "Case00_2", "Case01_1", "Case03_3"]) == \
            O.get_label(on, "prostate", parts, ["Case00_1", "Case03_0", "Case00_2", "Case01_1", "Case03_3"])
    with pytest.raises(NotImplementedError):
        get_label("cycle", "prostate", parts, groups)
    with pytest.raises(NotImplementedError):
        get_label("partition", "spleen", parts, groups)


def test_flip_batched_equals_per_sample_and_restores_rng_state():
    f = TensorRandomFlip(axis=[1, 2], threshold=0.8)
    x = torch.arange(6 * 2 * 4 * 5.).reshape(6, 2, 4, 5)
    random.seed(99)
    before = random.random()
    random.seed(99)
    with FixRandomSeed(1234):
        a = torch.stack([f(s) for s in x])
    assert random.random() == before  # state restored on exit
    with FixRandomSeed(1234):
        b = f.apply_batch(x)
    assert torch.equal(a, b)
    with FixRandomSeed(1234):
        c = f.apply_batch(x)  # cached plan, same result
    assert torch.equal(b, c)
    assert not torch.equal(a, x)


def test_meters_accumulate_tensors_and_floats():
    m = MeterInterface(default_focus="semi")
    with m.focus_on("hook"):
        m.register_meter("loss", AverageValueMeter())
        m["loss"].add(torch.tensor(2.0))
        m["loss"].add(torch.tensor(4.0))
        m["loss"].add(3.0)
        assert "loss" in m
    assert m.statistics()["hook"]["loss"]["mean"] == pytest.approx(3.0)
    m.reset()
    assert math.isnan(m.statistics()["hook"]["loss"]["mean"])


def test_hook_api_surface_and_combine():
    class H(EpocherHook):
        def __init__(self, v):
            super().__init__("h")
            self.v, self.calls = v, []

        def before_forward_pass(self, **kw):
            self.calls.append("bf")

        def __call__(self, **kw):
            return torch.tensor(self.v)

        def close(self):
            self.calls.append("close")

    a, b = H(1.0), H(2.5)
    c = CombineEpochHook(a, b)
    c.before_forward_pass()
    assert float(c()) == 3.5
    c.close()
    assert a.calls == ["bf", "close"]
    t = TrainerHook(hook_name="unique-name-for-test")
    assert list(t.parameters()) == []
    with pytest.raises(ValueError):
        TrainerHook(hook_name="unique-name-for-test")


def test_hook_factories_and_feature_until():
    model = UNet(input_dim=1, num_classes=4, max_channel=256)
    h1 = create_sp_infonce_hooks(model=model, feature_names=["Conv5", "Conv5"], weights=[1.0, 0.5],
                                 contrast_ons=["partition", "patient"], begin_values=3, end_values=70, mode="soft",
                                 max_epoch=80, correct_grad=True)
    assert feature_until_from_hooks(h1) == "Conv5"
    assert sum(p.numel() for p in h1.parameters()) == 2 * 131584
    keys = list(h1.state_dict().keys())
    assert "_hooks.0._projector._header.2.weight" in keys and "_hooks.1._projector._header.4.bias" in keys
    e1 = h1()  # epoch 0: gamma = begin value
    assert h1._hooks[0]._criterion.age_param == 3.0
    e1.close()
    h1()
    assert h1._hooks[0]._criterion.age_param == pytest.approx(3 + 67 * math.sqrt(1 / 80))
    h2 = create_infonce_hooks(model=model, feature_names="Conv4", weights=1.0, contrast_ons="self")
    assert feature_until_from_hooks(h2) == "Conv4"
    assert h2._hooks[0]._projector._header[2].in_features == 128
    # decoder feature: the dense head with its default (10, 10) pooling (hooks/infonce.py:73-76,101-106) ...
    h3 = create_infonce_hooks(model=model, feature_names="Up_conv3", weights=1.0, contrast_ons="self")
    assert feature_until_from_hooks(h3) == "Up_conv3"
    assert type(h3._hooks[0]._projector).__name__ == "DenseProjectionHead"
    assert tuple(h3._hooks[0]._projector._spatial_size) == (10, 10)
    assert "_hooks.0._projector._projector.0.weight" in h3.state_dict() and type(h3._hooks[0]()).__name__ == "_INFONCEDenseHook"
    assert feature_until_from_hooks(h2, h3) == "Up_conv3"
    # ... which has no self-paced form in the reference (its SP hook always hands out the encoder-style epoch hook)
    with pytest.raises(NotImplementedError):
        create_sp_infonce_hooks(model=model, feature_names="Up_conv3", weights=1.0, contrast_ons="self", begin_values=3,
                                end_values=70, mode="soft", max_epoch=80)


def test_unet_state_dict_keys_are_the_reference_keys():
    m = UNet(input_dim=1, num_classes=4, max_channel=256)
    keys = set(m.state_dict().keys())
    assert keys == set(O.init_unet_state(1, 4, 256).keys())
    n_enc = sum(p.numel() for n, p in m.named_parameters() if n.startswith("_Conv"))
    assert n_enc == 1179472  # encoder parameter count
    assert sum(p.numel() for p in m.parameters()) == 2160180  # full UNet (fine-tune DDP message 8.64 MB)


def test_feature_extractor_tap_semantics():
    m = UNet(input_dim=1, num_classes=4)
    ext = SingleFeatureExtractor(m, "Conv5")
    ext.bind()
    assert len(m._Conv5._forward_hooks) == 1
    with pytest.raises(RuntimeError):
        ext.feature()
    ext.set_enable(True)
    fake = torch.zeros(2, 3)
    for _ in range(4):
        ext._feature_extractor(None, None, fake)
    assert ext.feature().shape == (8, 3)
    with pytest.raises(RuntimeError):  # 5th un-cleared pass (arch/hook.py:25-27)
        ext._feature_extractor(None, None, fake)
    ext.clear()
    ext.remove()
    assert len(m._Conv5._forward_hooks) == 0


def test_warmup_cosine_schedule():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=5e-7)
    s = WarmupCosine(opt, max_epoch=80, warmup_max=10, multiplier=400)
    lrs = []
    for _ in range(80):
        lrs.append(opt.param_groups[0]["lr"])
        s.step()
    assert lrs[0] == pytest.approx(5e-7)
    assert lrs[10] == pytest.approx(5e-7 * 400)
    assert lrs[45] == pytest.approx(1e-7 + (2e-4 - 1e-7) * 0.5, rel=1e-6)
    assert all(a >= b for a, b in zip(lrs[10:], lrs[11:]))
    sd = s.state_dict()
    s2 = WarmupCosine(torch.optim.SGD([p], lr=5e-7), max_epoch=80, warmup_max=10, multiplier=400)
    s2.load_state_dict(sd)
    assert s2.epoch == 80


def test_gradient_sinks_are_armed_once_per_step_on_cpu():
    """ddp.FlatParams.zero_grad arms one sink per parameter (its slice of the flat bucket); functional.take_grad_sink hands it
    out once, and a gradient written there and returned as a fresh view is adopted by autograd without a copy."""
    from spcl_amd import ddp
    from spcl_amd.functional import take_grad_sink
    lin = torch.nn.Linear(4, 3)
    flat = ddp.FlatParams(lin.parameters())
    assert take_grad_sink(lin.weight) is None  # nothing armed yet
    flat.zero_grad()
    s = take_grad_sink(lin.weight)
    assert s is not None and s.data_ptr() == flat.views[0].data_ptr() and s.shape == lin.weight.shape
    assert take_grad_sink(lin.weight) is None  # one backward per arming
    assert take_grad_sink(lin.bias, needed=False) is None and take_grad_sink(lin.bias) is not None
    assert take_grad_sink(None) is None

    class WriteIntoSink(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w):
            ctx.sink = take_grad_sink(w)
            return w.sum()

        @staticmethod
        def backward(ctx, g):
            ctx.sink.fill_(3.0)
            return ctx.sink.view(ctx.sink.shape)

    flat.zero_grad()
    WriteIntoSink.apply(lin.weight).backward()
    assert lin.weight.grad.data_ptr() == flat.views[0].data_ptr()
    # the bias received no gradient: an optimizer over the flat parameter would move it anyway (torch.optim skips such
    # parameters), so gather refuses unless the zeros are asked for
    with pytest.raises(RuntimeError, match="received no gradient"):
        flat.gather_grads()
    lin2 = torch.nn.Linear(4, 3)
    flat2 = ddp.FlatParams(lin2.parameters(), allow_missing_grads=True)
    flat2.zero_grad()
    WriteIntoSink.apply(lin2.weight).backward()
    g = flat2.gather_grads()
    assert torch.equal(g[:12], torch.full((12,), 3.0)) and torch.equal(g[12:], torch.zeros(3))


def test_meter_batching_falls_back_to_immediate_adds_off_gpu():
    from spcl_amd.contrastyou import meters as M
    m = M.AverageValueMeter()
    M.begin_batch()
    m.add(torch.tensor(2.0))  # a CPU tensor is not batched (the one-launch kernel is a GPU path)
    m.add(4.0)
    M.flush_batch()
    assert m.summary()["mean"] == 3.0


def test_radam_host_coefficients_are_torch_s_formulas():
    """optim.radam_coefficients (the scalars a staged step uploads instead of launching the coefficient kernel): torch.optim.RAdam's
    bias corrections and rectification (torch/optim/radam.py _single_tensor_radam) for the step counts where the rectified branch
    switches on, beta^t by repeated squaring against ``**``, the learning rate taken as the float32 the device holds"""
    import math

    import numpy as np
    from spcl_amd.optim import _ipow, radam_coefficients
    b1, b2, lr = 0.9, 0.999, 2e-3
    for t in (1, 2, 3, 5, 6, 7, 64, 1000, 12345, 10 ** 6):
        # (the rounding of each squaring is squared with it: up to ~t / 2 ulp -- 1e-10 relative at a million steps, far below
        # the float32 the coefficients are rounded to)
        assert abs(_ipow(b2, t) - b2 ** t) <= max(4, t) * np.finfo(np.float64).eps * b2 ** t
        c_m, c_u, flag, tt = radam_coefficients(t, lr, b1, b2)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        rho_inf = 2 / (1 - b2) - 1
        rho_t = rho_inf - 2 * t * b2 ** t / bc2
        assert tt == float(t) and flag == (1.0 if rho_t > 5 else 0.0)
        np.testing.assert_allclose(c_m, float(np.float32(lr)) / bc1, rtol=1e-12)
        if rho_t > 5:
            rect = math.sqrt((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t))
            np.testing.assert_allclose(c_u, rect * math.sqrt(bc2), rtol=1e-10)
        else:
            assert c_u == 0.0
    assert radam_coefficients(5, lr, b1, b2)[2] == 0.0 and radam_coefficients(6, lr, b1, b2)[2] == 1.0  # rho_t crosses 5 at t = 6


def test_global_average_side_output_is_used_only_while_the_tensor_is_unchanged():
    """functional._gap_of: the [N, C] rows a block left on its activation are handed to the projector only for that very tensor
    object, with its version counter unmoved and matching shapes"""
    from spcl_amd.functional import _gap_of
    feat = torch.zeros(4, 8, 3, 3)
    gap = torch.zeros(4, 8)
    assert _gap_of(feat, 4, 8) is None
    feat._spcl_gap = (gap, feat._version)
    assert _gap_of(feat, 4, 8) is gap
    assert _gap_of(feat[:, :, :, :], 4, 8) is None     # a view is another object: it does not carry the attribute
    assert _gap_of(feat, 4, 16) is None and _gap_of(feat, 2, 8) is None
    feat.add_(1.0)                                       # written in place: the average is stale
    assert _gap_of(feat, 4, 8) is None
