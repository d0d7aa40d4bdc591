"""What one dependent kernel launch costs on this stack: N tiny kernels back to back, eager and replayed from a hipGraph."""
import torch, time
x = torch.zeros(64, device="cuda")
def body(n):
    for _ in range(n):
        x.add_(1.0)
def timed(fn, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (1, 10, 100, 400):
    body(n); torch.cuda.synchronize()
    t_e = timed(lambda: body(n))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(3)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(n)
    g.replay(); torch.cuda.synchronize()
    t_g = timed(g.replay)
    print(f"n={n}: eager {t_e:.1f} us ({t_e/n:.2f}/kernel), graph {t_g:.1f} us ({t_g/n:.2f}/kernel)")
