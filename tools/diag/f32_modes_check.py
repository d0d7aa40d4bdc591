"""f32 convolution kernels, split vs exact mode vs fp64, over small / odd shapes (GPU box):  python tools/diag/f32_modes_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from tests import test_gpu_kernels as K

n = K._n()
shapes = [(8, 32, 16, 8, 8), (8, 16, 16, 8, 8), (8, 32, 32, 4, 4), (8, 64, 32, 4, 4), (8, 16, 16, 16, 16), (8, 32, 16, 16, 16),
          (8, 16, 32, 8, 8), (8, 32, 64, 4, 4), (8, 64, 128, 2, 2), (8, 16, 16, 4, 4), (8, 32, 16, 4, 4),
          (8, 128, 64, 4, 4), (8, 128, 128, 2, 2), (8, 64, 64, 4, 4), (8, 128, 64, 8, 8), (8, 64, 32, 16, 16), (8, 32, 16, 32, 32),
          (8, 64, 32, 8, 8), (8, 32, 32, 16, 16), (8, 16, 16, 32, 32), (8, 128, 128, 4, 4), (2, 256, 128, 28, 28)]
for (N, ci, co, H, W) in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    dy = torch.randn(N, co, H, W, generator=g)
    sc, sh = torch.randn(ci, generator=g), torch.randn(ci, generator=g) * 0.3
    act = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    ref_f = F.conv2d(act.double(), w.double(), None, 1, 1)
    ref_d = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    ref_w = torch.nn.grad.conv2d_weight(act.double(), (co, ci, 3, 3), dy.double(), 1, 1)
    out = []
    for mode in (1, 0):
        n.call("spcl_conv_set_f32_split", mode)
        xs, dys = K.nhwc(x, torch.float32), K.nhwc(dy, torch.float32)
        wp0, wp1 = K.pack(n, w, 0, torch.float32), K.pack(n, w, 1, torch.float32)
        scd, shd = sc.cuda(), sh.cuda()
        y, _ = K.conv(n, xs, torch.float32, N, H, W, ci, ci, co, wp0, 1, scd, shd)
        dx, _ = K.conv(n, dys, torch.float32, N, H, W, co, co, ci, wp1, 0)
        ws = torch.empty(n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, ci, co) // 4, device="cuda")
        dw = torch.empty(co, ci, 3, 3, device="cuda")
        n.call("spcl_conv3x3_wgrad", n.ptr(xs), n.ptr(dys), n.dtype_code(torch.float32), N, H, W, ci, ci, ci, co, co, 1,
               n.ptr(scd), n.ptr(shd), n.ptr(ws), n.ptr(dw), n.stream())
        out.append((K.relerr(y.permute(0, 3, 1, 2).cpu(), ref_f), K.relerr(dx.permute(0, 3, 1, 2).cpu(), ref_d),
                    K.relerr(dw.cpu(), ref_w)))
    n.call("spcl_conv_set_f32_split", 1)
    print((N, ci, co, H, W), "split fwd %.1e dgrad %.1e wgrad %.1e | exact fwd %.1e dgrad %.1e wgrad %.1e" % (out[0] + out[1]), flush=True)
