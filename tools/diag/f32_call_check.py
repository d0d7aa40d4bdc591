"""every f32 convolution / weight-gradient call of the dense-hook fp32 test, run under BOTH f32 modes on the same inputs:
prints the calls whose results differ by more than 1e-4 (relative to the largest element)"""
import inspect, os, re, sys, textwrap
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.conftest  # noqa
import torch
import spcl_amd  # noqa
from spcl_amd import functional as Fn, native as n
from tests import test_gpu_round2_heads as M

def both(fn, name):
    def wrapped(*a, **k):
        out = fn(*a, **k)
        n.call("spcl_conv_set_f32_split", 0)
        ref = fn(*a, **k)
        n.call("spcl_conv_set_f32_split", 1)
        if isinstance(out, tuple) and len(out) > 1 and torch.is_tensor(out[1]) and hasattr(out[1], "ntiles"):
            cs = out[0].shape[-1]
            sa, sb = out[1][:out[1].ntiles * 3 * cs].view(-1, 3, cs).double(), ref[1][:out[1].ntiles * 3 * cs].view(-1, 3, cs).double()
            for comp, nm2 in enumerate(("count", "mean", "M2")):
                e = float((sa[:, comp] - sb[:, comp]).abs().max() / sb[:, comp].abs().max().clamp_min(1e-30))
                if e > 1e-4 or e != e:
                    print("BAD stats", nm2, "%.2e" % e, tuple(out[0].shape), "ntiles", out[1].ntiles, flush=True)
        o, r = (out[0] if isinstance(out, tuple) else out), (ref[0] if isinstance(ref, tuple) else ref)
        if torch.is_tensor(o) and o.dtype == torch.float32:
            err = float((o.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-30))
            desc = [tuple(v.shape) if torch.is_tensor(v) else v for v in a if torch.is_tensor(v) or isinstance(v, int)]
            print(("BAD " if err > 1e-4 or err != err else "ok  ") + name, "%.2e" % err, desc, flush=True)
        return out
    return wrapped

for nm in ("_conv", "_wgrad", "_conv_cat", "_wgrad_up2") if len(sys.argv) < 2 else sys.argv[1].split(","):
    setattr(Fn, nm, both(getattr(Fn, nm), nm))
src = textwrap.dedent(inspect.getsource(M.test_dense_infonce_hook_step_vs_oracle_fp32))
src = re.sub(r"assert rel\(p\.grad\.cpu\(\)\.numpy\(\), osd\[k\]\.grad\.numpy\(\)\) < 5e-3, .*", "print(k, '%.2e' % rel(p.grad.cpu().numpy(), osd[k].grad.numpy()))", src)
src = src.replace("def test_dense_infonce_hook_step_vs_oracle_fp32(", "def run(")
ns = dict(M.__dict__)
exec(src, ns)
try:
    ns["run"]()
except AssertionError as e:
    print("assert:", str(e)[:200])
