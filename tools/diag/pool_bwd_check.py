"""the unpooling backward through LeakyReLU (spcl_adaptive_avgpool2d_backward_act) at a training step's sizes against torch in float64"""
import sys, torch
sys.path.insert(0, "/root/repo")
import spcl_amd
from spcl_amd import native as _n
torch.manual_seed(0)
for (N, H, W, C, oh, ow) in [(60, 56, 56, 256, 10, 10), (8, 56, 56, 256, 10, 10), (60, 56, 56, 64, 10, 10), (60, 40, 40, 256, 10, 10), (2, 56, 56, 256, 10, 10)]:
    h = torch.randn(N, H, W, C, device="cuda")
    dhp = torch.randn(N, oh, ow, C, device="cuda")
    dpre = torch.empty(N, H, W, C, device="cuda")
    _n.call("spcl_adaptive_avgpool2d_backward_act", _n.ptr(dhp), _n.ptr(h), _n.dtype_code(torch.float32), N, H, W, C, oh, ow, _n.ptr(dpre), _n.stream())
    x = torch.zeros(N, C, H, W, device="cuda", dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.adaptive_avg_pool2d(x, (oh, ow))
    y.backward(dhp.permute(0, 3, 1, 2).double())
    ref = x.grad.permute(0, 2, 3, 1) * torch.where(h > 0, 1.0, 0.01).double()
    d = (dpre.double() - ref).abs()
    bad = (d > 1e-5).nonzero()
    print((N, H, W, C), "max err", float(d.max()), "bad", len(bad), bad[:5].tolist() if len(bad) else "")
