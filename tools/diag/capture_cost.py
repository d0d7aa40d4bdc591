"""Where the capture step of a (second and later) epoch goes: times around the pieces of StepGraph._capture."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-extras", "--no-roofline", "--no-cpu-baseline"]
args = bench.parse()
torch.cuda.set_device(0)
step, epocher, _ = bench.build_step(args, torch.device("cuda", 0), 0, 1)
from spcl_amd import stepgraph as sg  # noqa: E402

T = {}


def timed(name, fn):
    def w(*a, **k):  # (host time only: a device synchronisation inside a capture is illegal)
        t = time.perf_counter()
        out = fn(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t
        T["_order"] = T.get("_order", []) + [(name, time.perf_counter())]
        return out
    return w


sg._gc_settle = timed("gc_settle", sg._gc_settle)
torch.cuda.CUDAGraph.capture_begin = timed("capture_begin", torch.cuda.CUDAGraph.capture_begin)
torch.cuda.CUDAGraph.capture_end = timed("capture_end (instantiate)", torch.cuda.CUDAGraph.capture_end)
for rep in range(4):
    epocher._step_graph = None  # what a new epocher starts with (the stage and pair stay: smaller than a real new epoch)
    epocher.stage = None
    T.clear()
    ts = []
    for i in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    print(f"rep {rep}: steps {[round(x * 1e3, 2) for x in ts]} ms; inside the capture step: "
          + ", ".join(f"{k} {v * 1e3:.2f}" for k, v in T.items() if k != "_order")
          + "; begin->end (the python pass of the step) "
          + str(round(1e3 * ([t for n, t in T["_order"] if "capture_end" in n][0] - [t for n, t in T["_order"] if "capture_begin" in n][0]
                             - T["capture_end (instantiate)"]), 2)), flush=True)
