"""Experiment: can a hipGraph capture be ENDED and a second one BEGUN from inside a backward hook (autograd's device
thread), so that a step is replayed as graph A | eager collective start | graph B?  Prints what happens; no product code."""
import sys
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
l1 = torch.nn.Linear(256, 256).to(dev)
l2 = torch.nn.Linear(256, 256).to(dev)
x = torch.randn(64, 256, device=dev)
side = torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "relaxed"


def step(hook):
    for p in list(l1.parameters()) + list(l2.parameters()):
        p.grad = None
    h = torch.relu(l1(x))
    if hook is not None:
        h.register_hook(lambda g: hook() and None)
    y = l2(h).square().mean()
    y.backward()
    return y


side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        ref = step(None)
    ref_g = [p.grad.clone() for p in l1.parameters()]
torch.cuda.synchronize()

ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
events = []


def switch():
    import threading
    events.append(threading.current_thread().name)
    ga.capture_end()
    events.append("ended A")
    gb.capture_begin(pool=ga.pool(), capture_error_mode=mode)
    events.append("began B")
    return None


try:
    with torch.cuda.stream(side):
        ga.capture_begin(capture_error_mode=mode)
        out = step(switch)
        gb.capture_end()
    torch.cuda.synchronize()
    print("capture ok", events)
    x.add_(1.0)
    with torch.cuda.stream(side):
        want = None
    ga.replay()
    gb.replay()
    torch.cuda.synchronize()
    got = [p.grad.clone() for p in l1.parameters()]
    with torch.cuda.stream(side):
        chk = step(None)
    torch.cuda.synchronize()
    ok = all(torch.equal(a, p.grad) for a, p in zip(got, l1.parameters()))
    print("replay equals eager on new input:", ok, float(out), float(chk))
except Exception as e:  # noqa: BLE001
    print("FAILED", type(e).__name__, e, events)
