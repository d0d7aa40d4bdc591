"""dense-hook fp32 test with the split mode on for the convolutions only / the weight gradients only"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.conftest  # noqa
import spcl_amd  # noqa
from spcl_amd import functional as Fn, native as n
which = sys.argv[1]  # "conv": split convs, exact wgrad; "wgrad": the reverse
def with_mode(fn, mode):
    def w(*a, **k):
        old = n.call("spcl_conv_get_f32_split")
        n.call("spcl_conv_set_f32_split", mode)
        try:
            return fn(*a, **k)
        finally:
            n.call("spcl_conv_set_f32_split", old)
    return w
for nm in ("_wgrad", "_wgrad_up2", "_wgrad_cat") if hasattr(Fn, "_wgrad_cat") else ("_wgrad", "_wgrad_up2"):
    setattr(Fn, nm, with_mode(getattr(Fn, nm), 0 if which == "conv" else 1))
sys.argv = ["dense_hook_errs.py", "1" if which == "conv" else "0"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense_hook_errs.py"), run_name="__main__")
