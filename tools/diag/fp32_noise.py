"""Diagnostic (GPU box): per-tensor distance of the HIP fp32 step and of the fp32 CPU oracle from the fp64 oracle,
and the per-block forward feature errors, at BASELINE configs[0] shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import spcl_oracle as O
from tests.test_gpu_configs import _step, _oracle, _relmax, _rell2

size, bs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (224, 8)
run = _step(size, bs, torch.float32, "partition", 1.0, 10.0, "acdc")
l32, o32, lv32, _ = _oracle(run, "partition", 1.0, 10.0, "acdc")
l64, o64, lv64, _ = _oracle(run, "partition", 1.0, 10.0, "acdc", dtype=torch.float64)
print("loss hip %.8f o32 %.8f o64 %.8f" % (run["loss"], l32, l64))
for k, p in run["net"].named_parameters():
    if p.requires_grad and o64[k].grad is not None:
        g = p.grad.cpu().numpy()
        print(f"{k:26s} hip-vs-64 max {_relmax(g, o64[k].grad.numpy()):.2e} L2 {_rell2(g, o64[k].grad.numpy()):.2e} | "
              f"o32-vs-64 max {_relmax(o32[k].grad.numpy(), o64[k].grad.numpy()):.2e} L2 {_rell2(o32[k].grad.numpy(), o64[k].grad.numpy()):.2e}")
# forward features per block
x = torch.cat([run["img"], run["x2"]], 0)
net = run["net"]
sd = run["sd"]
with torch.no_grad():
    for until in ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5"):
        net.load_state_dict(sd, strict=True)  # resets BN buffers
        net.train()
        y = net(x.cuda(), until=until).float().cpu().double()
        y32 = O.encoder_forward(x, {k: v.clone() for k, v in sd.items()}, until)
        y64 = O.encoder_forward(x.double(), {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, until)
        print(until, "hip-vs-64 L2 %.2e  o32-vs-64 L2 %.2e" % (_rell2(y.numpy(), y64.numpy()), _rell2(y32.numpy(), y64.numpy())))
