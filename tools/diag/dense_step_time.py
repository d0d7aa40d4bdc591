"""Decoder pre-training with the dense InfoNCE hook (main_pretrain_decoder.py: encoder frozen, Up5 .. tapped block
train), 30 slices of 224^2 per step: the step issued launch by launch and replayed from a hipGraph."""
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
from spcl_amd.semi_seg.hooks import create_infonce_hooks  # noqa: E402
from spcl_amd.synthetic import SyntheticPretrainLoader  # noqa: E402

dev = torch.device("cuda", 0)
feature = sys.argv[1] if len(sys.argv) > 1 else "Up_conv3"
for graph in (False, True):
    torch.manual_seed(5)
    net = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
    net.set_compute_dtype(torch.bfloat16)
    hook = create_infonce_hooks(model=net, feature_names=feature, weights=1.0, contrast_ons="partition", data_name="acdc").to(dev)
    for p in net.parameters():
        p.requires_grad_(False)
    names = ["Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2"]
    for name in names[:names.index(feature) + 1]:
        getattr(net, "_" + name).requires_grad_(True)
    flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
    opt = FusedRAdam([flat.param], lr=1e-4, weight_decay=1e-5)
    loader = SyntheticPretrainLoader(bs=30, size=224, device=dev, seed=1, resident=True, meta="acdc", pool=8)
    ep = PretrainDecoderEpocher(model=net, optimizer=opt, chain_dataloader=loader, num_batches=10 ** 9, device=dev,
                                inference_until=feature, flat_params=flat, graph=graph)
    ep.add_hooks([hook()])
    net.train()
    with ep.meters.focus_on(ep.meter_focus):
        for _ in range(8):
            ep.step(next(loader))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            ep.step(next(loader))
        torch.cuda.synchronize()
    sg = ep._step_graph
    print(f"{feature} graph={graph} captured={bool(sg and sg.captured)}: {(time.perf_counter() - t0) * 10:.3f} ms per step", flush=True)
