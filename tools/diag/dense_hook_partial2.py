"""dense-hook fp32 test: split mode for forward convolutions only / for dgrad convolutions only / by image size"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.conftest  # noqa
import spcl_amd  # noqa
from spcl_amd import functional as Fn, native as n
which = sys.argv[1]
_conv = Fn._conv
def conv(x_store, dt_code, dtype, N, H, W, cin_s, cin_k, cout_s, wp, in_mode, scale, shift, want_stats):
    if which == "fwd":
        m = 1 if want_stats else 0
    elif which == "dgrad":
        m = 0 if want_stats else 1
    elif which.startswith("H"):
        m = 1 if H == int(which[1:]) else 0
    elif which.startswith("noise"):
        m = 0
    elif which.startswith("k"):
        m = 1 if cin_k == int(which[1:]) else 0
    if which.startswith("noise"):
        import torch
        y, st = _conv(x_store, dt_code, dtype, N, H, W, cin_s, cin_k, cout_s, wp, in_mode, scale, shift, want_stats)
        if H == 32 and want_stats:
            g = torch.Generator(device="cuda").manual_seed(5)
            y.mul_(1.0 + float(which[5:]) * torch.randn(y.shape, generator=g, device="cuda"))
        return y, st
    n.call("spcl_conv_set_f32_split", m)
    try:
        return _conv(x_store, dt_code, dtype, N, H, W, cin_s, cin_k, cout_s, wp, in_mode, scale, shift, want_stats)
    finally:
        n.call("spcl_conv_set_f32_split", 0)
Fn._conv = conv
sys.argv = ["dense_hook_errs.py", "0"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense_hook_errs.py"), run_name="__main__")
