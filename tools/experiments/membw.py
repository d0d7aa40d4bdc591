"""Experiment: plain HBM write / read / copy rates at the size of the materialised 4096^2 f32 logits (67 MB)."""
import torch
n = 4096 * 4096
bufs = [torch.empty(n, device="cuda") for _ in range(6)]  # rotate: 6 x 67 MB > the 256 MB infinity cache


def t(fn, it=30):
    for i in range(5):
        fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(it):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


print("fill  67 MB: %.1f us" % t(lambda i: bufs[i % 6].fill_(1.0)))
print("sum   67 MB: %.1f us" % t(lambda i: bufs[i % 6].sum()))
print("copy  67 MB: %.1f us" % t(lambda i: bufs[i % 6].copy_(bufs[(i + 3) % 6])))
one = bufs[0]
print("fill same buffer (cache-resident): %.1f us" % t(lambda i: one.fill_(1.0)))
print("sum  same buffer (cache-resident): %.1f us" % t(lambda i: one.sum()))
