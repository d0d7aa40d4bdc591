"""How fast is a validation pass?  EvalEpocher over L resident batches: wall clock per batch, and the GPU time of the same
launches (HIP events around the pass) -- the difference is the host issuing launches one by one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402,F401
from spcl_amd.contrastyou.losses.kl import KL_div  # noqa: E402
from spcl_amd.semi_seg.arch import UNet  # noqa: E402
from spcl_amd.semi_seg.epochers import EvalEpocher  # noqa: E402
from spcl_amd.synthetic import SyntheticLabeledLoader  # noqa: E402

dev = torch.device("cuda", 0)
bs, L = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 40
torch.manual_seed(3)
model = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
model.set_compute_dtype(torch.bfloat16)
loader = SyntheticLabeledLoader(bs=bs, size=224, device=dev, twice=False, length=L)
for rep in range(3):
    ep = EvalEpocher(model=model, loader=loader, sup_criterion=KL_div(), device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    stats = ep.run()
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"bs {bs}: wall {wall / L * 1e3:.3f} ms/batch  host-issue {host / L * 1e3:.3f}  events {e0.elapsed_time(e1) / L:.3f}  "
          f"dice {ep.get_score():.5f}", flush=True)
