"""dense-hook fp32 test errors with torch.empty replaced by zeros / by a NaN fill (uninitialised-read hunt)"""
import os, runpy, sys
import torch
mode = sys.argv[1]
_empty = torch.empty
def patched(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.device.type == "cuda":
        t.fill_(0.0 if mode == "zeros" else float("nan"))
    return t
torch.empty = patched
sys.argv = ["dense_hook_errs.py", "1"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense_hook_errs.py"), run_name="__main__")
