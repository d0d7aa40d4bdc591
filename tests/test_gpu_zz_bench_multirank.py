"""The N > 1 control flow of bench.py (rank environment, parameter / buffer broadcast, gradients written into the flat
bucket, split hipGraph capture with the collective between the two graphs, max-over-ranks timing, rank-0 JSON line)
exercised with TWO ranks on the ONE GPU of the test box: both ranks on cuda:0, gloo instead of RCCL
(SPCL_BENCH_ONE_DEVICE=1).  A self-test of the plumbing, not a measurement -- and never a gate for the parity tests:
the file sorts last, the run is bounded, its process group is killed as a whole, and a box on which two processes
cannot share the device in time is a SKIP, not a failure."""
import json
import os
import signal
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT_S = 150


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_device(tmp_path):
    env = dict(os.environ, SPCL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SPCL_BENCH_WATCHDOG_S="60",
               SPCL_BENCH_DDP_CHECK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--bs", "4", "--size", "64", "--no-extras"]
    so, se = open(tmp_path / "out.txt", "w+"), open(tmp_path / "err.txt", "w+")
    proc = subprocess.Popen(cmd, cwd=REPO, env=env, stdout=so, stderr=se, text=True, start_new_session=True)
    try:
        rc = proc.wait(timeout=LIMIT_S)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # launcher AND both workers: nothing may stay behind on cuda:0
        proc.wait()
        se.seek(0)
        pytest.skip(f"two processes did not share the device within {LIMIT_S} s: {se.read()[-1500:]}")
    finally:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    so.seek(0), se.seek(0)
    out, err = so.read(), se.read()
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["hipgraph"] == "epocher-split" and line["config"]["global_batch"] == 8
    assert line["value"] > 0 and line["final_meters"]["loss"] == line["final_meters"]["loss"]  # finite loss
    chk = line["ddp_check"]  # the collective left the mean of the two ranks' (different) gradients in the bucket
    assert chk["max_abs_diff_vs_mean_of_rank_gradients"] == 0.0 and chk["max_abs_diff_between_ranks"] > 0.0
    assert chk["grad_abs_max"] > 0.0
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
