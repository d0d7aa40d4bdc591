"""A pre-train step with the reference's three-position hook list (feature_names [Conv5, Up_conv3, Up_conv2], one global and two
dense InfoNCE hooks; 30 slices of 224^2, bf16): eager vs replayed -- equal losses, and the step time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402,F401
from spcl_amd import ddp  # noqa: E402
from spcl_amd.optim import FusedRAdam  # noqa: E402
from spcl_amd.semi_seg.arch import UNet  # noqa: E402
from spcl_amd.semi_seg.epochers import PretrainDecoderEpocher  # noqa: E402
from spcl_amd.semi_seg.hooks import create_infonce_hooks, feature_until_from_hooks  # noqa: E402
from spcl_amd.synthetic import SyntheticPretrainLoader  # noqa: E402

dev = torch.device("cuda", 0)
curves = {}
for graph in (False, True):
    torch.manual_seed(5)
    import random
    random.seed(11)
    net = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
    net.set_compute_dtype(torch.bfloat16)
    hook = create_infonce_hooks(model=net, feature_names=["Conv5", "Up_conv3", "Up_conv2"], weights=[1.0, 0.5, 0.25],
                                contrast_ons=["partition", "partition", "partition"], data_name="acdc").to(dev)
    until = feature_until_from_hooks(hook)
    for p in net.parameters():
        p.requires_grad_(False)
    with net.set_grad(True, start="Conv1", end=until):
        flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
        opt = FusedRAdam([flat.param], lr=1e-5, weight_decay=1e-5)
        loader = SyntheticPretrainLoader(bs=30, size=224, device=dev, seed=1, resident=True, meta="acdc", pool=8)
        ep = PretrainDecoderEpocher(model=net, optimizer=opt, chain_dataloader=loader, num_batches=10 ** 9, device=dev,
                                    inference_until=until, flat_params=flat, graph=graph)
        ep.add_hooks([hook()])
        net.train()
        losses = []
        with ep.meters.focus_on(ep.meter_focus):
            for _ in range(8):
                losses.append(ep.step(next(loader)).detach().clone())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                ep.step(next(loader))
            torch.cuda.synchronize()
        sg = ep._step_graph
        print(f"until {until} graph={graph} captured={bool(sg and sg.captured)}: {(time.perf_counter() - t0) * 20:.3f} ms per step",
              flush=True)
        curves[graph] = [float(x) for x in losses]
print("first eight losses equal:", curves[False] == curves[True], curves[True][:3])
