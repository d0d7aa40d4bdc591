"""per-parameter gradient errors of tests/test_gpu_round2_heads.py::test_dense_infonce_hook_step_vs_oracle_fp32 (prints, no asserts)"""
import inspect, os, re, sys, textwrap
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.conftest  # noqa
from tests import test_gpu_round2_heads as M
src = textwrap.dedent(inspect.getsource(M.test_dense_infonce_hook_step_vs_oracle_fp32))
src = re.sub(r"assert rel\(p\.grad\.cpu\(\)\.numpy\(\), osd\[k\]\.grad\.numpy\(\)\) < 5e-3, .*",
             "print(k, tuple(p.shape), '%.2e' % rel(p.grad.cpu().numpy(), osd[k].grad.numpy()))", src)
src = src.replace("def test_dense_infonce_hook_step_vs_oracle_fp32(", "def run(")
ns = dict(M.__dict__)
exec(src, ns)
if len(sys.argv) > 1:
    import spcl_amd  # noqa
    from spcl_amd import native
    native.call("spcl_conv_set_f32_split", int(sys.argv[1]))
import torch
for rep in range(int(os.environ.get("REPS", "1"))):
    torch.manual_seed(1234)
    torch.cuda.manual_seed_all(1234)
    print("---- rep", rep)
    try:
        ns["run"]()
    except AssertionError as e:
        print("assert:", str(e)[:300])
