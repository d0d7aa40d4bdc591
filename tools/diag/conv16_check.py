"""Row-mapped vs linear one-pass block-1 backward (csrc/conv16_bwd.hip) on a few shapes: this process runs ONE setting of
SPCL_CONV16_ROWMAP and saves / compares the results under gpurun_out/ (run under `timeout`: a faulting kernel hangs nothing
but the profiler).  usage: SPCL_CONV16_ROWMAP=0 python tools/diag/conv16_check.py save; SPCL_CONV16_ROWMAP=1 python ... cmp"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import spcl_amd  # noqa
from spcl_amd import functional as F_, native as _n

mode = sys.argv[1]
dtc = _n.dtype_code(torch.bfloat16)
os.makedirs("gpurun_out", exist_ok=True)
for (N, H, W) in [(2, 140, 154), (9, 224, 224), (1, 256, 256), (3, 140, 14), (64, 224, 224)]:
    gq = torch.Generator().manual_seed(11 + N)
    dy = torch.randn(N, H, W, 16, generator=gq).cuda().bfloat16()
    y2 = torch.randn(N, H, W, 16, generator=gq).cuda().bfloat16()
    w = torch.randn(16, 16, 3, 3, generator=gq).cuda() * 0.1
    st = [torch.randn(16, generator=gq).cuda() * 0.1, torch.rand(16, generator=gq).cuda() + 0.5,
          torch.rand(16, generator=gq).cuda() + 0.5, torch.randn(16, generator=gq).cuda() * 0.3]
    img = torch.rand(N, H, W, generator=gq).cuda()
    wpt = F_._pack(w, 1, dtc, torch.bfloat16)
    acorr = F_._image_autocorr(img.contiguous(), N, H, W)
    for wg in (False, True):
        dw, rows = F_._conv16_bwd_fused(dy, wpt, y2, st, img, dtc, N, H, W, 16, 16, 16, None, acorr=acorr if wg else None)
        torch.cuda.synchronize()
        tot = rows.view(11, 16, rows.wg).double().sum(2) if wg else rows.view(-1, 11, 16).double().sum(0)
        f = f"gpurun_out/c16_{N}_{H}_{W}_{int(wg)}.pt"
        if mode == "save":
            torch.save((dw.cpu(), tot.cpu()), f)
            print("saved", N, H, W, wg, flush=True)
        else:
            dw0, tot0 = torch.load(f)
            e1 = ((dw.cpu() - dw0).abs().max() / dw0.abs().max()).item()
            e2 = ((tot.cpu() - tot0).abs().max() / tot0.abs().max()).item()
            print(f"{N}x{H}x{W} wgrows={int(wg)}: dW max rel diff {e1:.2e} (equal: {torch.equal(dw.cpu(), dw0)}), row totals {e2:.2e}", flush=True)
