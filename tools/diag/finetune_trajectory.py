"""Fine-tune loss trajectory over many graph-replayed steps with the decoder's in-place paths on / off (environment switches are
read at import: run twice):   python tools/diag/finetune_trajectory.py 200 > a.txt;  SPCL_CONV_CAT=0 ... python ... > b.txt"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import spcl_amd  # noqa
from spcl_amd import ddp as _ddp
from spcl_amd.optim import FusedRAdam
from spcl_amd.contrastyou.losses.kl import KL_div
from spcl_amd.semi_seg.arch import UNet
from spcl_amd.semi_seg.epochers.finetune import FineTuneEpocher
from spcl_amd.synthetic import SyntheticLabeledLoader

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
m = UNet(input_dim=1, num_classes=4, max_channel=256).cuda()
m.set_compute_dtype(torch.bfloat16)
flat = _ddp.FlatParams([p for p in m.parameters() if p.requires_grad])
opt = FusedRAdam([flat.param], lr=1e-3)
loader = SyntheticLabeledLoader(bs=8, size=224, channels=1, num_classes=4, device="cuda", seed=5, pool=4)
ep = FineTuneEpocher(model=m, optimizer=opt, labeled_loader=loader, sup_criterion=KL_div(), num_batches=steps, device="cuda",
                     flat_params=flat, graph=True)
losses = []
with ep.meters.focus_on(ep.meter_focus):
    m.train()
    for i in range(steps):
        losses.append(ep.step(next(loader)).detach().clone())
torch.cuda.synchronize()
ls = torch.stack(losses).float().cpu()
assert torch.isfinite(ls).all()
for i in range(0, steps, max(1, steps // 20)):
    print(i, f"{float(ls[i]):.6f}")
print("last", f"{float(ls[-1]):.6f}", "mean last 10", f"{float(ls[-10:].mean()):.6f}")
