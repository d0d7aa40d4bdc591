"""split-mode f32 convolutions: the same call repeated, outputs compared bit for bit (race hunt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_kernels as K
n = K._n()
shapes = [(8, 16, 32, 16, 16, 0), (8, 32, 16, 16, 16, 0), (8, 16, 16, 16, 16, 1), (8, 32, 32, 8, 8, 1), (8, 32, 64, 8, 8, 0), (8, 64, 32, 8, 8, 0),
          (8, 64, 64, 4, 4, 1), (8, 128, 64, 4, 4, 0), (8, 64, 128, 4, 4, 0), (8, 128, 128, 2, 2, 1), (8, 16, 16, 32, 32, 1),
          (64, 16, 16, 224, 224, 1), (64, 64, 64, 56, 56, 1), (64, 256, 256, 14, 14, 1)]
for (N, ci, co, H, W, mode) in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    xs, wp = K.nhwc(x, torch.float32), K.pack(n, w, 0, torch.float32)
    sc, sh = torch.randn(ci, generator=g).cuda(), torch.randn(ci, generator=g).cuda()
    y0, st0 = K.conv(n, xs, torch.float32, N, H, W, ci, ci, co, wp, mode, sc, sh, stats=True)
    torch.cuda.synchronize()
    bad = 0
    reps = 200 if N * H * W < 100000 else 20
    for r in range(reps):
        y, st = K.conv(n, xs, torch.float32, N, H, W, ci, ci, co, wp, mode, sc, sh, stats=True)
        if not torch.equal(y, y0) or not torch.equal(st[:st0.ntiles * 3 * co], st0[:st0.ntiles * 3 * co]):
            bad += 1
    print((N, ci, co, H, W, mode), "nondeterministic runs:", bad, "of", reps, flush=True)
