"""Wall clock of one fine-tune EPOCH in the reference's own configuration (config/base.yaml: 200 training batches of 5
slices; then the validation and the test pass, one scan per batch -- semi_seg/data/creator.py:139-144 -- here 35 + 65
synthetic scans of 6 .. 18 slices at 224^2), with the validation passes issued launch by launch and replayed from hipGraphs."""
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
from spcl_amd.semi_seg.epochers.finetune import EvalEpocher, FineTuneEpocher  # noqa: E402
from spcl_amd.synthetic import SyntheticLabeledLoader  # noqa: E402

dev = torch.device("cuda", 0)


class Scans:
    def __init__(self, n_scans, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        lens = torch.randint(6, 19, (n_scans,), generator=torch.Generator().manual_seed(seed)).tolist()
        self.items = []
        for i, n in enumerate(lens):
            img = torch.rand((n, 1, 224, 224), device=dev, generator=g)
            tgt = torch.randint(0, 4, (n, 1, 224, 224), device=dev, generator=g)
            self.items.append(((img, tgt), [f"s{i}_{k}" for k in range(n)], ([0] * n, [f"s{i}"] * n)))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


torch.manual_seed(3)
model = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
model.set_compute_dtype(torch.bfloat16)
flat = ddp.FlatParams([p for p in model.parameters() if p.requires_grad])
opt = FusedRAdam([flat.param], lr=1e-5, weight_decay=1e-5)
train = SyntheticLabeledLoader(bs=5, size=224, device=dev, seed=77, pool=8)
val, test = Scans(35, 1), Scans(65, 2)
for graph in (False, True, False, True):
    for epoch in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train()
        ep = FineTuneEpocher(model=model, optimizer=opt, labeled_loader=train, sup_criterion=KL_div(verbose=False),
                             num_batches=200, device=dev, flat_params=flat)
        ep.run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for loader in (val, test):
            ev = EvalEpocher(model=model, loader=loader, sup_criterion=KL_div(verbose=False), device=dev, graph=graph)
            ev.run()
            ev.get_score()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"eval graphs {graph}  epoch {epoch}: train {t1 - t0:.3f} s  val+test {t2 - t1:.3f} s  total {t2 - t0:.3f} s",
              flush=True)
