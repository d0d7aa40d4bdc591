"""Where a 200-step fine-tune epoch (5 slices per batch) spends its wall clock: per-step host times of one FineTuneEpocher."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402,F401
from spcl_amd import ddp  # noqa: E402
from spcl_amd.contrastyou.losses.kl import KL_div  # noqa: E402
from spcl_amd.optim import FusedRAdam  # noqa: E402
from spcl_amd.semi_seg.arch import UNet  # noqa: E402
from spcl_amd.semi_seg.epochers.finetune import FineTuneEpocher  # noqa: E402
from spcl_amd.synthetic import SyntheticLabeledLoader  # noqa: E402

dev = torch.device("cuda", 0)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(3)
model = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
model.set_compute_dtype(torch.bfloat16)
flat = ddp.FlatParams([p for p in model.parameters() if p.requires_grad])
opt = FusedRAdam([flat.param], lr=1e-5, weight_decay=1e-5)
train = SyntheticLabeledLoader(bs=bs, size=224, device=dev, seed=77, pool=8)
for epoch in range(3):
    model.train()
    ep = FineTuneEpocher(model=model, optimizer=opt, labeled_loader=train, sup_criterion=KL_div(verbose=False),
                         num_batches=200, device=dev, flat_params=flat)
    ts = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ep.meters.focus_on(ep.meter_focus):
        for i in range(200):
            a = time.perf_counter()
            ep.step(next(train))
            if i < 4 or i == 199:
                torch.cuda.synchronize()
            ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    rest = sorted(ts[4:199])
    print(f"bs {bs} epoch {epoch}: total {tot * 1e3:.1f} ms; steps 0-3 (synchronised): {[round(t * 1e3, 2) for t in ts[:4]]} ms; "
          f"steps 4-198 host median {rest[len(rest) // 2] * 1e6:.0f} us; (total - first four) / 196 = "
          f"{(tot - sum(ts[:4])) / 196 * 1e3:.3f} ms", flush=True)
