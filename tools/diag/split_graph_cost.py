"""What the N > 1 control flow of a step costs on ONE GPU (no peers): the pre-train step of bench.py as

  1   one process: the whole step is one hipGraph                                  (the N = 1 bench)
  2   distributed, collectives stubbed out: compute graph | exchange | update graph
  3   the same with --ddp-overlap: compute head | early bucket | compute tail | exchange | update (StepGraph.cut)
  7   a REAL one-rank RCCL group (torch's own stream handling of the collective is in the number): two graphs
  9   one-rank RCCL group + --ddp-overlap: three graphs, the early bucket's all-reduce asynchronous
  11  one-rank RCCL group, SPCL_GRAPH_COLLECTIVE=1: the collective captured inside the step's one graph

Wall clock over K replays.  Round-6 numbers (profiles/r06_experiments/NOTES.md): 1: 0.989 ms, 7: 1.012 (1.071 while the
collective ran on a private communication stream between two stream waits), 9: 1.05, 11: 0.994."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def measure(mode, steps=200, warm=20):
    sys.argv = ["bench.py", "--no-extras", "--no-roofline", "--no-cpu-baseline"] + (["--ddp-overlap"] if mode in (3, 9) else [])
    args = bench.parse()
    import spcl_amd  # noqa: F401
    from spcl_amd import ddp
    import torch.distributed as dist
    if mode == 11:
        os.environ["SPCL_GRAPH_COLLECTIVE"] = "1"
    if mode in (7, 9, 11):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        ddp.is_distributed = lambda: True
    elif mode > 1:
        ddp.is_distributed = lambda: True
        dist.get_world_size = lambda group=None: 1

        class _Done:
            def wait(self):
                return True

        dist.all_reduce = lambda t, op=None, group=None, async_op=False: _Done() if async_op else None
        dist.broadcast = lambda *a, **k: None
    torch.cuda.set_device(0)
    step, epocher, _ = bench.build_step(args, torch.device("cuda", 0), 0, 1)
    for _ in range(6):
        step()
    sg = epocher._step_graph
    assert sg is not None and sg.captured, "not captured"
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"mode {mode}  graphs {len(sg._graphs)}  ms_per_step {dt:.4f}", flush=True)


if __name__ == "__main__":
    measure(int(sys.argv[1]))
