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


def _run_two_ranks(tmp_path, cmd, n=2, graphs="epocher-split"):
    env = dict(os.environ, SPCL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SPCL_BENCH_WATCHDOG_S="60",
               SPCL_BENCH_DDP_CHECK="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    so, se = open(tmp_path / "out.txt", "w+"), open(tmp_path / "err.txt", "w+")
    proc = subprocess.Popen(cmd, cwd=REPO, env=env, stdout=so, stderr=se, text=True, start_new_session=True)
    try:
        rc = proc.wait(timeout=LIMIT_S)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGTERM)  # bench.py's own parent forwards it to the launcher's process group
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
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
    assert line["n_gpus"] == n and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["hipgraph"] == graphs and line["config"]["global_batch"] == 4 * n
    assert line["value"] > 0 and line["final_meters"]["loss"] == line["final_meters"]["loss"]  # finite loss
    chk = line["ddp_check"]  # the collective left the mean of the ranks' (different) gradients in the bucket
    # (two addends have one order: exact; from three on the collective's order is its own)
    assert chk["max_abs_diff_vs_mean_of_rank_gradients"] <= (0.0 if n == 2 else 1e-6 * chk["grad_abs_max"])
    assert chk["max_abs_diff_between_ranks"] > 0.0
    assert chk["grad_abs_max"] > 0.0
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    mg = line["multi_gpu"]  # every rank's own step time and the collective alone (diagnosable on first contact)
    assert len(mg["per_rank_ms_per_step"]) == n and all(v > 0 for v in mg["per_rank_ms_per_step"])
    assert max(mg["per_rank_ms_per_step"]) <= line["ms_per_step"] * 1.001 and mg["allreduce_us"] > 0
    return line


BENCH_ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--bs", "4", "--size", "64", "--no-extras"]


def test_bench_two_ranks_on_one_device(tmp_path):
    """under the launcher, as the driver starts it"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + BENCH_ARGS
    _run_two_ranks(tmp_path, cmd)


def test_plain_bench_gpus_2_spawns_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no launcher and no WORLD_SIZE: bench.py starts torch.distributed.run as a child
    process itself (before any GPU call), forwards rank 0's line and the exit code -- it must not silently measure one
    GPU and print n_gpus: 1 (VERDICT r03 missing #1; reference seam: semi_seg/main_infonce.py:35,39)."""
    _run_two_ranks(tmp_path, [sys.executable, os.path.join(REPO, "bench.py")] + BENCH_ARGS)


def test_bench_four_ranks_on_one_device(tmp_path):
    """the same control flow at world size 4 (nothing beyond two ranks had ever run): four ranks share cuda:0"""
    args = [a if a != "2" or i != 1 else "4" for i, a in enumerate(BENCH_ARGS)]
    assert args[:2] == ["--gpus", "4"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + args
    _run_two_ranks(tmp_path, cmd, n=4)


def test_bench_two_ranks_overlapped_early_bucket_is_three_graphs(tmp_path):
    """``--ddp-overlap`` in graph mode: the compute graph is cut at the Conv3 | Conv2 boundary from inside
    backward (stepgraph.StepGraph.cut, called by ddp.FlatParams.reduce_early on autograd's thread), the early bucket's
    asynchronous all-reduce starts between the two compute graphs of every replay, the head's follows the second, then the
    update graph.  Same collective semantics: the bucket holds the mean of the ranks' gradients bit for bit, and the
    run's loss meter equals the one-bucket run's."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1"]
    (tmp_path / "a").mkdir(), (tmp_path / "b").mkdir()
    over = _run_two_ranks(tmp_path / "a", launcher + ["--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] +
                          BENCH_ARGS + ["--ddp-overlap"], graphs="epocher-split3")
    plain = _run_two_ranks(tmp_path / "b", launcher + ["--master-port", str(_free_port()),
                                                       os.path.join(REPO, "bench.py")] + BENCH_ARGS)
    a, b = over["final_meters"]["loss"], plain["final_meters"]["loss"]
    assert abs(a - b) <= 1e-5 * abs(b), (a, b)
