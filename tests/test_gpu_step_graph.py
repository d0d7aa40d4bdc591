"""The epocher's step replayed from a hipGraph (stepgraph.py) against the same steps launched eagerly: identical bits.

The product loop ``PretrainEncoderEpocher._run_pretrain`` (mirror of semi_seg/epochers/new_pretrain.py:52-89) captures its
step after two eager iterations; every later iteration refills the stage (label vectors, flip flags) from the new batch
and replays.  Fresh images, a fresh slice order (hence label vector) and a fresh flip seed every step."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(graph, dtype=torch.float32, sync_checks=False, cmax=128, size=32, three_hooks=False, seed=3):
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    torch.manual_seed(seed)
    net = UNet(input_dim=1, num_classes=4, max_channel=cmax).cuda()
    net.set_compute_dtype(dtype)
    if three_hooks:
        hook = create_sp_infonce_hooks(model=net, feature_names=["Conv5"] * 3, weights=[1.0, 0.5, 0.25],
                                       contrast_ons=["partition", "patient", "self"], begin_values=8.0, end_values=8.0,
                                       mode="soft", max_epoch=10, p=0.5, correct_grad=True, data_name="prostate",
                                       sync_checks=sync_checks).cuda()
    else:
        hook = create_sp_infonce_hooks(model=net, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                       begin_values=9.0, end_values=9.0, mode="soft", max_epoch=10, p=0.5,
                                       correct_grad=True, data_name="acdc", sync_checks=sync_checks).cuda()
    for name in net.decoder_names:
        getattr(net, "_" + name).requires_grad_(False)
    flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
    opt = FusedRAdam([flat.param], lr=2e-3, weight_decay=1e-5)
    ep = PretrainEncoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([]), num_batches=100, device="cuda",
                                inference_until="Conv5", flat_params=flat, graph=graph)
    ep.add_hooks([hook()])
    net.train()
    return net, hook, flat, opt, ep


def _batches(steps, bs, size, meta="acdc", seed=17):
    from spcl_amd.synthetic import acdc_like_meta, prostate_like_meta
    g = torch.Generator().manual_seed(seed)
    out = []
    tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long).cuda()
    for k in range(steps):
        a, b = torch.rand(bs, 1, size, size, generator=g).cuda(), torch.rand(bs, 1, size, size, generator=g).cuda()
        fn, part, grp = (prostate_like_meta(bs, 4, shift=3 * k) if meta == "prostate" else acdc_like_meta(bs, shift=5 * k))
        out.append(((a, b, tgt, tgt), fn, (part, grp)))
    return out


def _run(ep, batches, seeds=None):
    curve = []
    random.seed(99)
    with ep.meters.focus_on(ep.meter_focus):
        for k, batch in enumerate(batches):
            loss = ep.step(batch, seed=None if seeds is None else seeds[k])
            curve.append(loss.detach().clone())
    torch.cuda.synchronize()
    return [float(c) for c in curve]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_steps_equal_eager_steps_bit_for_bit(dtype):
    steps, bs = 7, 12
    res = {}
    for graph in (False, True):
        net, hook, flat, opt, ep = _setup(graph, dtype)
        curve = _run(ep, _batches(steps, bs, 32))
        sg = ep._step_graph
        if graph:
            assert sg is not None and sg.captured and not sg.failed and sg.replays == steps - 2, (sg.replays,)
        else:
            assert sg is None
        stats = ep.meters.statistics()
        res[graph] = (curve, flat.data.clone(), {k: v.clone() for k, v in net.state_dict().items()},
                      {f"{g}/{k}": v["mean"] for g, ms in stats.items() for k, v in ms.items()},
                      [float(s) for s in opt.state[flat.param]["step"].reshape(1)])
    ce, cg = res[False][0], res[True][0]
    assert ce == cg, (ce, cg)
    assert len(set(ce)) == steps  # the batches (and so the losses) really differ from step to step
    assert torch.equal(res[False][1], res[True][1])
    for k, v in res[False][2].items():
        assert torch.equal(v, res[True][2][k]), k
    assert res[False][4] == res[True][4] == [float(steps)]
    for k, v in res[False][3].items():  # meters: loss, sp_weight, age_param (a python float re-applied per replay), reg_loss
        np.testing.assert_allclose(v, res[True][3][k], rtol=1e-6, err_msg=k)
    assert {k.split("/")[-1] for k in res[True][3]} >= {"loss", "sp_weight", "age_param", "reg_loss"}


def test_graphed_three_hooks_and_sync_checks():
    """three meta-label hooks on one feature (batched projection + batched losses, row N4) through the graph, with the
    reference's per-step assertions on (``sync_checks=True``: checked after each replay)"""
    steps, bs = 6, 8
    res = {}
    for graph in (False, True):
        net, hook, flat, opt, ep = _setup(graph, torch.float32, sync_checks=True, three_hooks=True)
        curve = _run(ep, _batches(steps, bs, 32, meta="prostate"))
        if graph:
            assert ep._step_graph.captured and ep._step_graph.replays == steps - 2
            crit = hook._hooks[1]._criterion
            assert 0.0 < crit.downgrade_ratio <= 1.0  # reads the captured result block after the last replay
        res[graph] = (curve, flat.data.clone())
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1])


def test_replay_reports_nan_like_the_eager_step():
    """``RuntimeError(loss)`` on NaN (contrast_loss3.py:203-204) survives the graph.  A replayed step is checked without
    draining the queue (its result block is copied to pinned memory behind the replay, the PREVIOUS step's copy is looked at):
    the error of step k is raised by step k + 1 -- or, for the last step, when the epoch's hooks close."""
    for how in ("next step", "close"):
        net, hook, flat, opt, ep = _setup(True, torch.float32, sync_checks=True)
        batches = _batches(6, 8, 32)
        _run(ep, batches[:4])
        assert ep._step_graph.captured
        with torch.no_grad():
            flat.data[-300:] = float("nan")  # the projector's last bias (a NaN in a conv weight dies in the next ReLU's fmax)
        with ep.meters.focus_on(ep.meter_focus):
            ep.step(batches[4])  # the step that earns the error: enqueued, not waited for
        with pytest.raises((RuntimeError, AssertionError)):
            if how == "close":
                ep.close_hooks()
            else:
                with ep.meters.focus_on(ep.meter_focus):
                    ep.step(batches[5])
        if how == "next step":
            # a staged step that raised drops the optimizer's host mirror of the step count (it advances at fill time, before
            # it is known whether the update launch follows: ADVICE r05); the device counter is the authority -- both replays ran
            assert not opt._step_host and int(opt.state[flat.param]["step"].item()) == 6
        if how == "close":
            ep._hooks = []  # (already closed)


def test_ragged_batch_runs_eagerly_and_graph_survives():
    net, hook, flat, opt, ep = _setup(True, torch.float32)
    full = _batches(5, 12, 32)
    small = _batches(1, 6, 32, seed=5)
    _run(ep, full[:4])
    sg = ep._step_graph
    n = sg.replays
    _run(ep, small)          # other shape: the old eager path
    assert sg.replays == n and sg.captured
    _run(ep, full[4:])
    assert sg.replays == n + 1


def test_new_epoch_recaptures_with_the_new_age_parameter():
    """gamma is baked into the capture; the next epoch's hook (new gamma) belongs to a new epocher and a new graph"""
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    res = {}
    for graph in (False, True):
        net, hook, flat, opt, ep = _setup(graph, torch.float32)
        hook._hooks[0]._scheduler.begin_value, hook._hooks[0]._scheduler.end_value = 4.0, 40.0
        curves = []
        for epoch in range(2):
            ep = PretrainEncoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([]), num_batches=100,
                                        device="cuda", inference_until="Conv5", flat_params=flat, graph=graph)
            ep.add_hooks([hook()])
            curves += _run(ep, _batches(4, 12, 32, seed=30 + epoch))
            ep.close_hooks()
        res[graph] = (curves, flat.data.clone())
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1])


def test_stage_bytes_roundtrip():
    """spcl_stage_bytes: host bytes -> device block, also beyond one launch's 3 584 bytes"""
    import ctypes
    from spcl_amd import native as _n
    for nbytes in (4, 64, 3584, 3588, 9000):
        src = np.random.RandomState(nbytes).randint(0, 256, size=nbytes, dtype=np.uint8)
        dst = torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda")
        _n.call("spcl_stage_bytes", _n.ptr(dst), src.ctypes.data_as(ctypes.c_void_p), nbytes, _n.stream())
        got = dst.cpu().numpy()
        assert np.array_equal(got[:nbytes], src) and not got[nbytes:].any()


def test_trainer_loop_uses_the_graph():
    """PretrainEncoderTrainer.start_training -> epocher.run -> _run_pretrain: the loop a user gets through the seam"""
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.semi_seg.trainers import PretrainEncoderTrainer
    from spcl_amd.synthetic import SyntheticPretrainLoader
    torch.manual_seed(5)
    net = UNet(input_dim=1, num_classes=4, max_channel=128).cuda()
    loader = SyntheticPretrainLoader(bs=9, size=32, device="cuda", seed=3, resident=True, pool=3)
    seen = []

    class Spy(PretrainEncoderTrainer):
        def _create_tra_epoch(self):
            ep = super()._create_tra_epoch()
            seen.append(ep)
            return ep

    with net.set_grad(False, start="Conv5", include_start=False):
        tr = Spy(model=net, chain_dataloader=loader, max_epoch=3, num_batches=6, device="cuda", lr=1e-3)
        tr.register_hooks(create_sp_infonce_hooks(model=net, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                                  begin_values=5.0, end_values=50.0, mode="soft", max_epoch=3, p=0.5,
                                                  correct_grad=True, data_name="acdc", sync_checks=False))
        tr.forward_until = "Conv5"
        tr.init()
        hist = tr.start_training()
    assert len(hist) == 2 and len(seen) == 2
    for ep in seen:
        assert ep._step_graph is not None and ep._step_graph.captured and ep._step_graph.replays == 4
    for h in hist:
        assert np.isfinite(h["semi"]["reg_loss"]["mean"])


def test_finetune_graphed_steps_equal_eager_steps():
    """FineTuneEpocher (new_epocher.py:260-283: full UNet, softmax + KL_div, Dice of the training batch): replayed steps
    equal eager steps bit for bit, and the Dice meter receives every step's counts"""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import FineTuneEpocher
    steps, bs, size = 6, 6, 32
    g = torch.Generator().manual_seed(8)
    batches = []
    for k in range(steps):
        img = torch.rand(bs, 1, size, size, generator=g).cuda()
        tgt = torch.randint(0, 4, (bs, 1, size, size), generator=g).cuda()
        groups = [f"patient{(i + k) % 3:03d}_00" for i in range(bs)]
        batches.append(((img, img, tgt, tgt), [f"f{i}" for i in range(bs)], (["0"] * bs, groups)))
    res = {}
    for graph in (False, True):
        torch.manual_seed(4)
        net = UNet(input_dim=1, num_classes=4, max_channel=128).cuda()
        flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad])
        opt = FusedRAdam([flat.param], lr=1e-3, weight_decay=1e-5)
        ep = FineTuneEpocher(model=net, optimizer=opt, labeled_loader=iter([]), sup_criterion=KL_div(), num_batches=steps,
                             device="cuda", flat_params=flat, graph=graph)
        net.train()
        curve = []
        with ep.meters.focus_on(ep.meter_focus):
            for b in batches:
                curve.append(float(ep.step(b).detach()))
        if graph:
            assert ep._step_graph.captured and ep._step_graph.replays == steps - 2
        stats = ep.meters.statistics()["semi"]
        res[graph] = (curve, flat.data.clone(), stats)
    assert res[False][0] == res[True][0]
    assert len(set(res[False][0])) == steps
    assert torch.equal(res[False][1], res[True][1])
    assert res[False][2]["sup_dice"] == res[True][2]["sup_dice"]
    np.testing.assert_allclose(res[False][2]["sup_loss"]["mean"], res[True][2]["sup_loss"]["mean"], rtol=1e-6)


def _setup_dense(graph, dtype):
    """decoder pre-training (main_pretrain_decoder.py:66-69: the encoder frozen, the decoder up to the tapped block trains)
    with the DENSE InfoNCE hook on Up_conv3 (semi_seg/hooks/infonce.py:201-241)"""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import PretrainDecoderEpocher
    from spcl_amd.semi_seg.hooks import create_infonce_hooks, feature_until_from_hooks
    torch.manual_seed(5)
    net = UNet(input_dim=1, num_classes=4, max_channel=128).cuda()
    net.set_compute_dtype(dtype)
    hook = create_infonce_hooks(model=net, feature_names="Up_conv3", weights=0.5, contrast_ons="partition",
                                data_name="acdc").cuda()
    assert feature_until_from_hooks(hook) == "Up_conv3"
    assert type(hook._hooks[0]._projector).__name__ == "DenseProjectionHead"
    for p in net.parameters():
        p.requires_grad_(False)
    for name in ("Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3"):
        getattr(net, "_" + name).requires_grad_(True)
    flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
    opt = FusedRAdam([flat.param], lr=2e-3, weight_decay=1e-5)
    ep = PretrainDecoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([]), num_batches=100, device="cuda",
                                inference_until="Up_conv3", flat_params=flat, graph=graph)
    ep.add_hooks([hook()])
    net.train()
    return net, hook, flat, opt, ep


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dense_hook_steps_replay_bit_for_bit(dtype):
    """The dense decoder hook (round 6: replayable).  Its host-side draws -- the sample-wise feature flips and the five points
    of every slice, numpy's RNG under the step's seed -- reach the captured launches through the stage; the replayed steps
    equal the eager steps (``apply_batch`` + ``region_extractor``'s advanced indexing) bit for bit, seeds fresh every step."""
    steps, bs = 7, 8
    res = {}
    for graph in (False, True):
        net, hook, flat, opt, ep = _setup_dense(graph, dtype)
        curve = _run(ep, _batches(steps, bs, 64))
        sg = ep._step_graph
        if graph:
            assert sg is not None and sg.captured and not sg.failed and sg.replays == steps - 2, (sg and sg.replays,)
        else:
            assert sg is None
        res[graph] = (curve, flat.data.clone(), {k: v.clone() for k, v in net.state_dict().items()})
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert len(set(res[False][0])) == steps
    assert torch.equal(res[False][1], res[True][1])
    for k, v in res[False][2].items():
        assert torch.equal(v, res[True][2][k]), k


def test_replayed_steps_with_contract_checks_do_not_drain_the_queue():
    """``sync_checks=True`` -- what hooks built from a config run with -- must not cost a device readback per step (round 6: it
    did, and halved the real trainer's throughput): with torch's synchronisation debug mode set to ``error`` the replayed
    steps run through (the check reads the PREVIOUS step's pinned copy behind an event), while the eager form of the same
    check (``criterion.check()``: ``tolist()`` of a device tensor) is caught by that mode."""
    net, hook, flat, opt, ep = _setup(True, torch.float32, sync_checks=True)
    batches = _batches(8, 8, 32)
    _run(ep, batches[:4])
    assert ep._step_graph.captured
    crit = hook._hooks[0]._criterion  # (the trainer-level hook's criterion: the epoch hook shares it)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        with ep.meters.focus_on(ep.meter_focus):
            for b in batches[4:]:
                ep.step(b)
        crit._host_out = None
        with pytest.raises(RuntimeError):
            crit.check()  # the per-step readback this test guards against
    finally:
        torch.cuda.set_sync_debug_mode("default")
    ep.close_hooks()  # (the last step's pending check)
    assert ep._step_graph.replays == 6
