"""The validation pass replayed from hipGraphs (``EvalEpocher``, one graph per batch shape, kept on the model across
epochers): same loss and Dice as the eager pass bit for bit -- across scans of different lengths, after the weights changed
in place between two epochs, and after the parameters MOVED (a ``FlatParams`` built later: the captured graphs read the old
storage and must be dropped).  Reference loop: semi_seg/epochers/new_epocher.py:77-97 (``EvalEpocher._run``), one scan per
batch as ``ScanBatchSampler`` hands them over (semi_seg/data/creator.py:139-144)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Scans:
    """finite, re-iterable loader: one 'scan' (a batch of its own slice count) per item, single-transform format"""

    def __init__(self, lengths, size=64, seed=5, classes=4):
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.items = []
        for i, n in enumerate(lengths):
            img = torch.rand((n, 1, size, size), device="cuda", generator=g)
            coarse = torch.randint(0, classes, (n, 1, size // 8, size // 8), device="cuda", generator=g)
            tgt = torch.nn.functional.interpolate(coarse.float(), size=(size, size), mode="nearest").long()
            names = [f"scan{i:02d}_{k:02d}" for k in range(n)]
            self.items.append(((img, tgt), names, ([0] * n, [f"scan{i:02d}"] * n)))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def _model(dtype):
    import spcl_amd  # noqa: F401
    from spcl_amd.semi_seg.arch import UNet
    torch.manual_seed(21)
    m = UNet(input_dim=1, num_classes=4, max_channel=128).cuda()
    m.set_compute_dtype(dtype)
    return m


def _eval(model, loader, graph):
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.semi_seg.epochers.finetune import EvalEpocher
    ep = EvalEpocher(model=model, loader=loader, sup_criterion=KL_div(verbose=False), device="cuda", graph=graph)
    stats = ep.run()
    with ep.meters.focus_on("eval"):
        dice = ep.meters["dice"]
        inter = torch.cat(dice._intersections, 0).cpu()
        union = torch.cat(dice._unions, 0).cpu()
        names = list(dice._group_names)
        loss = float(ep.meters["loss"].summary()["mean"]) if hasattr(ep.meters["loss"], "summary") else None
    return stats, inter, union, names, loss, ep.get_score()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_replayed_validation_equals_eager_validation(dtype):
    from spcl_amd.semi_seg.epochers.finetune import _EvalGraphs
    model = _model(dtype)
    loader = _Scans([6, 9, 6, 11, 9, 6, 11, 9, 6])  # three shapes, each seen at least three times: eager, capture, replay
    ref = _eval(model, loader, graph=False)
    got = _eval(model, loader, graph=True)
    graphs = _EvalGraphs.of(model)
    assert len(graphs.entries) == 3 and all(e["graph"] is not None for e in graphs.entries.values())
    assert torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2]) and ref[3] == got[3]
    assert ref[4] == got[4] and ref[5] == got[5]

    # second epoch: the weights changed IN PLACE (an optimizer step, load_state_dict): every batch is a replay now
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.05)
    ref2 = _eval(model, loader, graph=False)
    got2 = _eval(model, loader, graph=True)
    assert _EvalGraphs.of(model) is graphs and len(graphs.entries) == 3
    assert not torch.equal(ref[1], ref2[1]) or ref[4] != ref2[4]  # (the pass really sees the new weights)
    assert torch.equal(ref2[1], got2[1]) and torch.equal(ref2[2], got2[2]) and ref2[4] == got2[4] and ref2[5] == got2[5]

    # the parameters MOVE into one flat tensor (what a trainer built after this point does): stale graphs are dropped
    from spcl_amd import ddp
    sig = graphs.signature
    ddp.FlatParams([p for p in model.parameters() if p.requires_grad])
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.9)
    ref3 = _eval(model, loader, graph=False)
    got3 = _eval(model, loader, graph=True)
    assert graphs.signature != sig
    assert torch.equal(ref3[1], got3[1]) and torch.equal(ref3[2], got3[2]) and ref3[4] == got3[4] and ref3[5] == got3[5]


def test_replayed_validation_batches_do_not_synchronise():
    """a validation pass whose batches are all replays issues no synchronising call before its statistics are read (torch's
    sync-debug mode set to ``error`` around the batch loop)"""
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.semi_seg.epochers.finetune import EvalEpocher
    model = _model(torch.bfloat16)
    loader = _Scans([6, 9, 6, 9, 6, 9])
    _eval(model, loader, graph=True)  # eager, capture, replay of both shapes
    ep = EvalEpocher(model=model, loader=loader, sup_criterion=KL_div(verbose=False), device="cuda", graph=True)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        with ep.meters.focus_on(ep.meter_focus):
            ep._run()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert 0.0 <= ep.get_score() <= 1.0
