"""Ablation timing of csrc/conv16_bwd.hip at the benchmark's size (64 x 224 x 224): which phase costs what.
    python tools/diag/conv16_phases.py        (needs an MI355X; the SPCL_CONV16_DBG bits of Bwd16Args.dbg and the stamps act
    only in a library built with -DSPCL_CONV16_DBG_BUILD=1 / -DSPCL_CONV16_STAMPS_BUILD=1; QUICK=1: the full kernel only)"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import spcl_amd  # noqa
from spcl_amd import functional as F_, native as _n

N, H, W = 64, 224, 224
dtc = _n.dtype_code(torch.bfloat16)
g = torch.Generator().manual_seed(0)
dy = torch.randn(N, H, W, 16, generator=g).cuda().bfloat16()
y2 = torch.randn(N, H, W, 16, generator=g).cuda().bfloat16()
w = torch.randn(16, 16, 3, 3, generator=g).cuda() * 0.1
st = [torch.zeros(16).cuda(), torch.ones(16).cuda(), torch.ones(16).cuda(), torch.zeros(16).cuda()]
img = torch.rand(N, H, W, generator=g).cuda()
wpt = F_._pack(w, 1, dtc, torch.bfloat16)


def timed(fn, reps=5, inner=20):
    """median over `reps` of the mean GPU time of `inner` back-to-back calls (buffers preallocated: the host stays ahead)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


nt = _n.call("spcl_conv_stat_rows", dtc, N, H, W, 16, 16)
rows = torch.empty(nt * 11 * 16, dtype=torch.float32, device="cuda")
ws = torch.empty(_n.call("spcl_conv16_bwd_fused_splits", N, H, W) * 9 * 256, dtype=torch.float32, device="cuda")
dw = torch.empty(16, 16, 3, 3, device="cuda")
stream = _n.stream()


def fused():
    _n.call("spcl_conv16_bwd_fused", _n.ptr(dy), dtc, N, H, W, _n.ptr(wpt), _n.ptr(y2), _n.ptr(st[2]), _n.ptr(st[3]),
            _n.ptr(st[0]), _n.ptr(img), _n.ptr(rows), _n.ptr(ws), _n.ptr(dw), 16, 16, None, None, 0, None, stream)


for dbg in ([0] if os.environ.get("QUICK") else [0, 1, 2, 4, 8, 16, 1 | 2, 4 | 8, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 16]):
    os.environ["SPCL_CONV16_DBG"] = str(dbg)
    print(f"dbg {dbg:2d}: {timed(fused):7.1f} us (kernel + its reduce launch)")
os.environ["SPCL_CONV16_DBG"] = "0"
os.environ["SPCL_CONV16_STAMPS"] = "1"  # (prints only in a -DSPCL_CONV16_STAMPS_BUILD=1 build)
fused()
del os.environ["SPCL_CONV16_STAMPS"]
if os.environ.get("QUICK"):
    sys.exit(0)
wsw = torch.empty(_n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, 16, 16) // 4, dtype=torch.float32, device="cuda")
t = timed(lambda: _n.call("spcl_conv3x3_wgrad", _n.ptr(y2), _n.ptr(dy), dtc, N, H, W, 16, 16, 16, 16, 16, 1, _n.ptr(st[2]),
                          _n.ptr(st[3]), _n.ptr(wsw), _n.ptr(dw), stream))
print(f"stand-alone wgrad (+ reduce): {t:7.1f} us")
t = timed(lambda: _n.call("spcl_conv3x3_dgrad_bnstats_image", _n.ptr(dy), dtc, N, H, W, 16, 16, _n.ptr(wpt), None, _n.ptr(y2),
                          _n.ptr(st[2]), _n.ptr(st[3]), _n.ptr(st[0]), _n.ptr(img), _n.ptr(rows), stream))
print(f"stand-alone dgrad (MODE 4, no g): {t:7.1f} us")
