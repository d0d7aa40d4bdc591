"""one layer of the >= 64-channel convolution, a few launches (for rocprofv3 --pmc): gemm_one.py H Cin Cout [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spcl_amd  # noqa
from spcl_amd import native as n
H, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N, dtc, dtype = 64, 1, torch.bfloat16
x = torch.randn(N, H, H, ci, device="cuda").to(dtype)
w = torch.randn(co, ci, 3, 3, device="cuda") / 10
sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
wp = torch.empty(n.call("spcl_conv_packed_elems", ci, co, 0, dtc), dtype=dtype, device="cuda")
n.call("spcl_conv_pack_weights", n.ptr(w), ci, co, 0, dtc, n.ptr(wp), n.stream())
y = torch.empty(N, H, H, co, dtype=dtype, device="cuda")
nt = n.call("spcl_conv_stat_rows", dtc, N, H, H, ci, co)
st = torch.empty(n.call("spcl_bn_stats_elems", nt, co), device="cuda")
for _ in range(5):
    n.call("spcl_conv3x3_forward", n.ptr(x), dtc, N, H, H, ci, ci, co, n.ptr(wp), mode, n.ptr(sc), n.ptr(sh), n.ptr(y),
           n.ptr(st), n.stream())
torch.cuda.synchronize()
