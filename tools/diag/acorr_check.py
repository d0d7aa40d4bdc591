import os, sys, torch
sys.path.insert(0, "/root/repo")
import spcl_amd  # noqa
from spcl_amd import functional as F_, native as _n
dtc = _n.dtype_code(torch.bfloat16)
for (N, H, W) in [(3, 224, 224), (2, 140, 154), (64, 224, 224)]:
    g = torch.Generator().manual_seed(N)
    img = torch.rand(N, H, W, generator=g).cuda()
    w = torch.randn(16, 1, 3, 3, generator=g).cuda() * 0.3
    wp = F_._pack(w, 0, dtc, torch.bfloat16)
    rows = F_._acorr_in_conv_rows(dtc, N, H, W)
    print(N, H, W, "rows", rows)
    ref = F_._image_autocorr(img.contiguous(), N, H, W).double().sum(0)
    y0, s0 = F_._conv(img.view(N, H, W, 1), dtc, torch.bfloat16, N, H, W, 1, 16, 16, wp, 2, None, None, True)
    if rows:
        y1, s1, ac = F_._conv_image_acorr(img.view(N, H, W, 1), dtc, torch.bfloat16, N, H, W, 1, 16, wp, True, rows)
        tot = ac.double().sum(0)
        print("  y equal", torch.equal(y0, y1), "stats equal", torch.equal(s0[:s0.ntiles*48], s1[:s1.ntiles*48]),
              "acorr rel diff", ((tot - ref).abs().max() / ref.abs().max()).item(), "pad", tot[54:].abs().max().item())
