"""Micro-benchmark of individual C-ABI kernels at the BASELINE layer shapes (HIP-event timed, isolated launches)."""
from ctypes import c_float
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spcl_amd  # noqa
from spcl_amd import native as n

LAYERS = [  # (name, H, Cin, Cout)
    ("C1a", 224, 1, 16), ("C1b", 224, 16, 16), ("C2a", 112, 16, 32), ("C2b", 112, 32, 32), ("C3a", 56, 32, 64),
    ("C3b", 56, 64, 64), ("C4a", 28, 64, 128), ("C4b", 28, 128, 128), ("C5a", 14, 128, 256), ("C5b", 14, 256, 256)]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    N = int(os.environ.get("N", "64"))
    dtype = torch.bfloat16
    dtc, es = 1, 2
    which = sys.argv[1:] or ["fwd", "dgrad", "wgrad", "bnfwd", "bnbwd"]
    for name, H, ci, co in LAYERS:
        W = H
        cs_i, cs_o = max(16, ci), co
        img = ci < 16
        x = (torch.rand(N, H, W, ci, device="cuda") if img else
             torch.randn(N, H, W, cs_i, device="cuda").to(dtype))
        dy = torch.randn(N, H, W, cs_o, device="cuda").to(dtype)
        w = torch.randn(co, ci, 3, 3, device="cuda") / 10
        sc = torch.rand(cs_i, device="cuda") + 0.5
        sh = torch.randn(cs_i, device="cuda") * 0.1
        px = N * H * W
        line = f"{name} {H}x{H} {ci}->{co}: "
        if "fwd" in which:
            wp = torch.empty(n.call("spcl_conv_packed_elems", ci, co, 0, dtc), dtype=dtype, device="cuda")
            n.call("spcl_conv_pack_weights", n.ptr(w), ci, co, 0, dtc, n.ptr(wp), n.stream())
            y = torch.empty(N, H, W, cs_o, dtype=dtype, device="cuda")
            nt_ = n.call("spcl_conv_stat_rows", dtc, N, H, W, 16 if img else cs_i, cs_o)
            st = torch.empty(n.call("spcl_bn_stats_elems", nt_, cs_o), device="cuda")
            mode = 2 if img else 1
            t = timeit(lambda: n.call("spcl_conv3x3_forward", n.ptr(x), dtc, N, H, W, ci if img else cs_i, 16 if img else cs_i,
                                      cs_o, n.ptr(wp), mode, n.ptr(sc), n.ptr(sh), n.ptr(y), n.ptr(st), n.stream()))
            byts = px * ((ci * 4 if img else cs_i * es) + cs_o * es)
            fl = 2.0 * px * 9 * ci * co
            line += f"fwd {t:7.1f}us ({byts / t / 1e3:6.0f} GB/s, {fl / t / 1e6:6.1f} TF) | "
            gam, bet = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda")
            o4 = torch.empty(4, cs_o, device="cuda")
            t = timeit(lambda: n.call("spcl_bn_finalize", n.ptr(st), nt_, co, cs_o, n.ptr(gam), n.ptr(bet), c_float(0.1),
                                      c_float(1e-5), None, None, None, n.ptr(o4[0]), n.ptr(o4[1]), n.ptr(o4[2]),
                                      n.ptr(o4[3]), n.stream()))
            line += f"bnfin {t:5.1f}us | "
        if "dgrad" in which and not img:
            wp = torch.empty(n.call("spcl_conv_packed_elems", ci, co, 1, dtc), dtype=dtype, device="cuda")
            n.call("spcl_conv_pack_weights", n.ptr(w), ci, co, 1, dtc, n.ptr(wp), n.stream())
            dx = torch.empty(N, H, W, cs_i, dtype=dtype, device="cuda")
            t = timeit(lambda: n.call("spcl_conv3x3_forward", n.ptr(dy), dtc, N, H, W, cs_o, cs_o, cs_i, n.ptr(wp), 0, None,
                                      None, n.ptr(dx), None, n.stream()))
            byts = px * (cs_i + cs_o) * es
            line += f"dgrad {t:7.1f}us ({byts / t / 1e3:6.0f} GB/s) | "
        if "wgrad" in which:
            ws = torch.empty(n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, 16 if img else cs_i, cs_o) // 4, device="cuda")
            dw = torch.empty(co, ci, 3, 3, device="cuda")
            mode = 2 if img else 1
            t = timeit(lambda: n.call("spcl_conv3x3_wgrad", n.ptr(x), n.ptr(dy), dtc, N, H, W, ci, ci if img else cs_i,
                                      16 if img else cs_i, co, cs_o, mode, n.ptr(sc), n.ptr(sh), n.ptr(ws), n.ptr(dw), n.stream()))
            byts = px * ((ci * 4 if img else cs_i * es) + cs_o * es)
            fl = 2.0 * px * 9 * ci * co
            line += f"wgrad {t:7.1f}us ({byts / t / 1e3:6.0f} GB/s, {fl / t / 1e6:6.1f} TF) | "
        if "bnfwd" in which:
            scale = torch.rand(cs_o, device="cuda") + 0.5
            shift = torch.randn(cs_o, device="cuda") * 0.1
            pool = torch.empty(N, H // 2, W // 2, cs_o, dtype=dtype, device="cuda")
            t = timeit(lambda: n.call("spcl_bnrelu_pool_forward", n.ptr(dy), dtc, N, H, W, cs_o, n.ptr(scale), n.ptr(shift),
                                      None, n.ptr(pool), n.stream()))
            byts = px * cs_o * es * 1.25
            line += f"bnfwd {t:6.1f}us ({byts / t / 1e3:6.0f} GB/s) | "
        if "bnbwd" in which:
            scale = torch.rand(cs_o, device="cuda") + 0.5
            shift = torch.randn(cs_o, device="cuda") * 0.1
            mean = torch.zeros(cs_o, device="cuda")
            istd = torch.ones(cs_o, device="cuda")
            dpool = torch.randn(N, H // 2, W // 2, cs_o, device="cuda").to(dtype)
            ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs_o) // 4, device="cuda")
            dg, db = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
            dyo = torch.empty(N, H, W, cs_o, dtype=dtype, device="cuda")
            t = timeit(lambda: n.call("spcl_bnrelu_pool_backward", n.ptr(dy), None, n.ptr(dpool), dtc, N, H, W, co, cs_o,
                                      n.ptr(mean), n.ptr(istd), n.ptr(scale), n.ptr(shift), 1, n.ptr(ws), n.ptr(dg),
                                      n.ptr(db), n.ptr(dyo), n.stream()))
            byts = px * cs_o * es * (2 * 1.25 + 1)
            line += f"bnbwd(pool) {t:6.1f}us ({byts / t / 1e3:6.0f} GB/s) | "
            t = timeit(lambda: n.call("spcl_bnrelu_pool_backward", n.ptr(dy), n.ptr(x if not img else dy), None, dtc, N, H, W,
                                      co, cs_o, n.ptr(mean), n.ptr(istd), n.ptr(scale), n.ptr(shift), 1, n.ptr(ws),
                                      n.ptr(dg), n.ptr(db), n.ptr(dyo), n.stream())) if cs_i == cs_o or img else float("nan")
            byts = px * cs_o * es * 5
            line += f"bnbwd(lin) {t:6.1f}us ({byts / t / 1e3:6.0f} GB/s)"
        if "imgwg" in which and img and ci == 1:
            scale = torch.rand(cs_o, device="cuda") + 0.5
            shift = torch.randn(cs_o, device="cuda") * 0.1
            mean, istd = torch.zeros(cs_o, device="cuda"), torch.ones(cs_o, device="cuda")
            g_ = torch.randn(N, H, W, cs_o, device="cuda").to(dtype)
            ws = torch.empty(n.call("spcl_bnrelu_image_wgrad_workspace_bytes", N, H, W, cs_o) // 4, device="cuda")
            dg, db = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
            dw = torch.empty(co, 1, 3, 3, device="cuda")
            t = timeit(lambda: n.call("spcl_bnrelu_backward_image_wgrad", n.ptr(dy), n.ptr(g_), n.ptr(x), dtc, N, H, W, co,
                                      cs_o, n.ptr(mean), n.ptr(istd), n.ptr(scale), n.ptr(shift), 1, n.ptr(ws), n.ptr(dg),
                                      n.ptr(db), n.ptr(dw), n.stream()))
            byts = px * cs_o * es * 4
            line += f"bnbwd+imgwgrad {t:6.1f}us ({byts / t / 1e3:6.0f} GB/s)"
        print(line)


if __name__ == "__main__":
    main()
