"""Micro-benchmark of the batched weight-gradient launch (csrc/wgrad_gemm.hip) on the five >=64-channel encoder layers
of the benchmark step (N=64 images of 224x224), all together and one layer at a time.  Uses the library's own kernel
timer (HIP events on the launch stream around every kernel) so that the two kernels of a call are reported apart."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spcl_amd  # noqa
from spcl_amd import native as n

LAYERS = [("C3b", 56, 64, 64, 1), ("C4a", 28, 64, 128, 0), ("C4b", 28, 128, 128, 1), ("C5a", 14, 128, 256, 0),
          ("C5b", 14, 256, 256, 1)]


def make(N, H, ci, co, mode, keep):
    x = torch.randn(N, H, H, ci, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, H, co, device="cuda").to(torch.bfloat16)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    dw = torch.zeros(co, ci, 3, 3, device="cuda")
    keep += [x, dy, sc, sh, dw]
    return n.WgradItem(x.data_ptr(), dy.data_ptr(), sc.data_ptr() if mode else None, sh.data_ptr() if mode else None,
                       dw.data_ptr(), N, H, H, ci, ci, co, co, mode)


def run(items, reps=20):
    arr = (n.WgradItem * len(items))(*items)
    ws = torch.empty(n.call("spcl_conv_wgrad_batched_workspace_bytes", arr, len(items)) // 4, device="cuda")
    for _ in range(3):
        n.call("spcl_conv3x3_wgrad_batched", arr, len(items), 0, n.ptr(ws), n.stream())
    torch.cuda.synchronize()
    n.call("spcl_profile_enable", 1)
    for _ in range(reps):
        n.call("spcl_conv3x3_wgrad_batched", arr, len(items), 0, n.ptr(ws), n.stream())
    torch.cuda.synchronize()
    cnt = n.call("spcl_profile_count")
    name = ctypes.create_string_buffer(256)
    us, by, fl = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
    acc = {}
    for i in range(cnt):
        n.call("spcl_profile_get", i, name, 256, ctypes.byref(us), ctypes.byref(by), ctypes.byref(fl))
        a = acc.setdefault(name.value.decode(), [0.0, 0, 0.0])
        a[0] += us.value
        a[1] += 1
        a[2] += fl.value
    n.call("spcl_profile_enable", 0)
    return {k: (v[0] / v[1], v[2] / v[1]) for k, v in acc.items()}


def main():
    N = int(os.environ.get("N", "64"))
    keep = []
    items = [make(N, H, ci, co, mode, keep) for _, H, ci, co, mode in LAYERS]
    total = 0.0
    for (name, H, ci, co, mode), it in zip(LAYERS, items):
        if os.environ.get("ONLY_BATCHED"):
            break
        r = run([it])
        fl = 2.0 * N * H * H * 9 * ci * co
        t = sum(v[0] for v in r.values())
        total += t
        print(f"{name} alone: " + ", ".join(f"{k.split('::')[-1].split('<')[0]} {v[0]:.1f} us" for k, v in r.items()) +
              f"  -> {fl / t / 1e6:.0f} TF")
    r = run(items)
    fl = sum(2.0 * N * H * H * 9 * ci * co for _, H, ci, co, _ in LAYERS)
    t = sum(v[0] for v in r.values())
    print("batched:  " + ", ".join(f"{k.split('::')[-1].split('<')[0]} {v[0]:.1f} us" for k, v in r.items()) +
          f"  -> {fl / t / 1e6:.0f} TF all-in ({fl / 1e9:.1f} GF in {t:.1f} us); one at a time: {total:.1f} us")


if __name__ == "__main__":
    main()
