"""native entry points the dense-hook fp32 test calls (unique names, in first-use order)"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.conftest  # noqa
import spcl_amd  # noqa
from spcl_amd import native as n
seen = []
_call = n.call
def call(name, *a):
    if name not in seen:
        seen.append(name)
    return _call(name, *a)
n.call = call
sys.argv = ["dense_hook_errs.py", "1"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense_hook_errs.py"), run_name="__main__")
print("CALLS:", " ".join(seen))
