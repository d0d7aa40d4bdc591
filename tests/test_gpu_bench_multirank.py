"""The N > 1 control flow of bench.py (rank environment, parameter / buffer broadcast, gradients written into the flat
bucket, split hipGraph capture with the collective between the two graphs, max-over-ranks timing, rank-0 JSON line)
exercised with TWO ranks on the ONE GPU of the test box: both ranks on cuda:0, gloo instead of RCCL
(SPCL_BENCH_ONE_DEVICE=1).  A self-test of the plumbing, not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, SPCL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--bs", "4"]  # default flags otherwise: the roofline pass runs on every rank
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["hipgraph"] == "split" and line["config"]["global_batch"] == 8
    assert line["value"] > 0 and line["final_meters"]["loss"] == line["final_meters"]["loss"]  # finite loss
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
