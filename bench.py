#!/usr/bin/env python3
"""Benchmark of the pre-train hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one rank per GPU over RCCL.  Under a launcher (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``:
RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process IS a rank; without one, ``python bench.py --gpus N`` starts that
launcher itself as a child process (spawn_ranks) and forwards its output and exit code.  WORLD_SIZE != N exits 2.

A "step" = one pass of the hot path over one synthetic batch, exactly the reference's pre-train iteration
(semi_seg/epochers/new_pretrain.py:53-89): two views of bs=32 slices (64 images of 1x224x224) -> UNet encoder to Conv5
-> forward-hook tap -> ProjectionHead(256,256,256) -> SelfPacedSupConLoss(soft, gamma 3->70, correct_grad) ->
backward -> (flat RCCL gradient all-reduce when N>1) -> RAdam step.  bf16 storage, fp32 accumulation/statistics/loss
(BASELINE.json configs[1]).  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline      the dominant kernel of the step -- one __global__ template with ALL its instantiations' launches together
                (conv3x3_fast_kernel: 17 launches, 42 % of the step) --, timed live with HIP events on its own stream in an
                instrumented pass right after the timed region; algorithmic bytes/FLOPs as defined in DESIGN.md
  roofline_symbol  the same for the single kernel SYMBOL with the largest total time (what rocprofv3 --stats ranks first)
  cpu_baseline  the CPU oracle's restatement of the same step ("port"), timed on this box's host cores (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3


class Watchdog:
    """N > 1 only: a rank whose peers stopped answering must EXIT non-zero instead of sitting in a collective.  A daemon
    thread watches a heartbeat the main thread touches between phases / steps; after ``limit`` seconds of silence it
    dumps every thread's stack and leaves with ``os._exit(3)`` (never a re-exec of a process that holds the GPU)."""

    def __init__(self, limit):
        import threading
        self.limit, self.t, self.phase, self.on = float(limit), time.monotonic(), "start", True
        threading.Thread(target=self._run, daemon=True).start()

    def beat(self, phase=None):
        self.t = time.monotonic()
        if phase is not None:
            self.phase = phase

    def stop(self):
        self.on = False

    def _run(self):
        import faulthandler
        while self.on:
            time.sleep(1.0)
            if self.on and time.monotonic() - self.t > self.limit:
                print(f"[bench] rank {os.environ.get('RANK', '0')}: no progress for {self.limit:.0f} s in phase "
                      f"'{self.phase}' -- a peer or a collective stalled; exiting 3", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(3)


class _NoWatchdog:
    def beat(self, phase=None):
        pass

    def stop(self):
        pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bs", type=int, default=32, help="slices per GPU (two views each)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly (default: the epocher replays its step from a hipGraph, stepgraph.py)")
    ap.add_argument("--pool", type=int, default=8,
                    help="distinct resident synthetic batches the loader cycles through (fresh images, slice order and "
                         "label vector every step; all in HBM before the timed region)")
    ap.add_argument("--fresh-rand", action="store_true",
                    help="draw every batch on the device inside the step (two torch.rand launches) instead of the pool")
    ap.add_argument("--ddp-overlap", action="store_true",
                    help="N>1: all-reduce the early gradient bucket (projector + Conv5..Conv3) from a backward hook while "
                         "Conv2..Conv1 are differentiated (ddp.enable_unet_overlap): the step is then THREE hipGraphs -- "
                         "forward + backward down to the Conv3 | Conv2 boundary, the rest of backward, the update -- with the "
                         "early collective started between the first two (stepgraph.StepGraph.cut)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (per-replay distribution, contrastive 4096x128 microbench)")
    ap.add_argument("--f32-exact", action="store_true",
                    help="--dtype fp32 only: multiply on v_mfma_f32_16x16x4_f32 (exact f32 products, 1/16 of the bf16 matrix "
                         "rate) instead of three bf16 pieces per operand (spcl_conv_set_f32_split(0))")
    ap.add_argument("--cpu-bs", type=int, default=8, help="batch of the CPU-oracle sample (bounded work)")
    ap.add_argument("--workload", default="pretrain", choices=["pretrain", "contrastive", "finetune", "prostate"],
                    help="pretrain = BASELINE configs[1] (the metric); prostate = configs[3] shape: 256x256, bs=64/GPU, "
                         "three self-paced hooks (partition, patient, self) sharing one encoder pass")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ step construction
def build_step(args, device, rank, world):
    if getattr(args, "f32_exact", False):
        import spcl_amd  # noqa: F401
        from spcl_amd import native as _native
        _native.call("spcl_conv_set_f32_split", 0)
    import spcl_amd
    from spcl_amd import ddp
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.synthetic import SyntheticPretrainLoader

    torch.manual_seed(10)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = UNet(input_dim=1, num_classes=4, max_channel=256, momentum=0.1).to(device)
    model.set_compute_dtype(dtype)
    prostate = getattr(args, "workload", "pretrain") == "prostate"
    if prostate:  # three meta-labels combined on the same feature (run_self_paced_acdc:61-70 shape, hooks/creator.py:102-124)
        hook = create_sp_infonce_hooks(model=model, feature_names=["Conv5"] * 3, weights=[1.0, 1.0, 1.0],
                                       contrast_ons=["partition", "patient", "self"], begin_values=3.0, end_values=70.0,
                                       mode="soft", max_epoch=80, p=0.5, correct_grad=True, data_name="prostate",
                                       sync_checks=False).to(device)
    else:
        hook = create_sp_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                       begin_values=3.0, end_values=70.0, mode="soft", max_epoch=80, p=0.5,
                                       correct_grad=True, data_name="acdc", sync_checks=False).to(device)
    for sub in hook._hooks:  # mid-schedule age parameter (epoch 40 of 80: gamma = 3 + 67*sqrt(0.5) = 50.4): at epoch 0
        sub._scheduler.epoch = 40  # gamma=3 < log(63) zeroes every weight at random init, i.e. a degenerate loss
    ddp.broadcast_state(model, hook)
    # freeze the decoder exactly as main_pretrain_encoder.py:69 does (set_grad(False, start="Conv5", include_start=False))
    for name in model.decoder_names:
        getattr(model, "_" + name).requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    hparams = list(hook.parameters())
    flat = ddp.FlatParams(params + hparams)  # one flat parameter + one flat gradient bucket (all-reduced when N>1)
    from spcl_amd.optim import FusedRAdam
    opt = FusedRAdam([flat.param], lr=5e-7 * 400, weight_decay=1e-5)  # torch.optim.RAdam semantics, HIP kernel
    loader = SyntheticPretrainLoader(bs=args.bs, size=args.size, device=device, seed=1234 + rank,
                                     resident=not getattr(args, "fresh_rand", False),
                                     meta="prostate" if prostate else "acdc", pool=getattr(args, "pool", 1))
    # the product loop's own step: PretrainEncoderEpocher.step captures itself in a hipGraph after two eager steps and
    # replays it from then on (labels / flip flags of every new batch through its stage, stepgraph.py)
    epocher = PretrainEncoderEpocher(model=model, optimizer=opt, chain_dataloader=loader, num_batches=10 ** 9,
                                     device=device, inference_until="Conv5", flat_params=flat,
                                     graph=not getattr(args, "no_graph", False))
    epocher.add_hooks([hook()])
    if getattr(args, "ddp_overlap", False):
        ddp.enable_unet_overlap(flat, model)
    model.train()
    nparams = sum(p.numel() for p in params + hparams)
    import random
    random.seed(4321 + rank)  # the per-step flip seeds (new_pretrain.py:54) are drawn from python's RNG

    def step():
        """one iteration of the epocher's loop on the loader's NEXT batch, with a freshly drawn flip seed"""
        with epocher.meters.focus_on(epocher.meter_focus):
            return epocher.step(next(loader))

    step.epocher, step.batch, step.flat, step.model = epocher, next(loader), flat, model
    return step, epocher, nparams


# ------------------------------------------------------------------------------------------------ roofline (live)
def _read_profile_log(lo, hi):
    """entries lo .. hi - 1 of the library's launch log as (kernel symbol, seconds, algorithmic bytes, FLOPs)"""
    import ctypes
    from spcl_amd import native
    name = ctypes.create_string_buffer(256)
    us, by, fl = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
    out = []
    for i in range(lo, hi):
        native.call("spcl_profile_get", i, name, 256, ctypes.byref(us), ctypes.byref(by), ctypes.byref(fl))
        out.append((name.value.decode(), us.value * 1e-6, by.value, fl.value))
    return out


def _pmc_table():
    """HBM bytes per launch by kernel symbol: NOT measured in this run -- read from the newest committed PMC passes of
    profiles/ (tools/pmc_traffic.py: separate rocprofv3 --pmc runs, 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"""
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic.json")))[-1]  # newest round
        return json.load(open(latest)), os.path.relpath(latest, REPO)
    except Exception:  # noqa: BLE001
        return {}, None


def _short_kernel_name(n):
    import re
    n = n.split("(")[0].replace("void ", "")
    return re.sub(r"\s+", "", n)


_PROFILER_ENV = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_PATH")
_PROFILER_ENV_PREFIXES = ("ROCPROFILER_", "ROCPROF_", "ROCP_")


def _under_profiler(env=None):
    """is a profiler's tool library loaded into THIS process (rocprofv3 -- python bench.py ...)?"""
    env = os.environ if env is None else env
    for k, v in env.items():
        if not v:
            continue
        if k == "LD_PRELOAD" and ("rocprof" in v or "roctx" in v or "rocprofiler" in v):
            return True
        if k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB") or k.startswith(("ROCPROFILER_", "ROCPROF_")):
            return True
    return False


def graph_kernel_times(args, first_kernel="flip_pair_stage"):
    """Per-launch kernel durations INSIDE the replayed step, by `tools/step_timeline.py`'s method: a CHILD process runs this
    same command (no extras / roofline / CPU baseline) under `rocprofv3 --kernel-trace`; its trace is cut into steps at every
    launch of the step's first kernel and the median duration of every launch position over the last 20 steps is returned as
    [(kernel name, seconds)].  HIP events recorded during a stream capture return no time on this stack (tried: the
    event-record nodes replay, hipEventElapsedTime fails), and the profiler cannot attach to a running process -- hence the
    child, which is a plain subprocess (the program after `--` is python itself: no exec of a process that holds the GPU).
    None when rocprofv3 is missing or the trace cannot be read."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if _under_profiler():
        # this process is itself being profiled: the child would inherit the preloaded tool library, which initialises the
        # GPU in the env -> python3 -> target chain of the rocprofv3 script -- the exec-after-GPU-init hop this pool refuses
        print("[bench] running under a profiler: no rocprofv3 child (pass --no-roofline there)", file=sys.stderr)
        return None
    out = tempfile.mkdtemp(prefix="spcl_bench_trace_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
           "--no-cpu-baseline", "--no-extras", "--no-roofline", "--steps", "30", "--warmup", "5", "--bs", str(args.bs),
           "--size", str(args.size), "--dtype", args.dtype, "--workload", args.workload, "--pool", str(args.pool)]
    if getattr(args, "f32_exact", False):
        cmd.append("--f32-exact")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for k in [k for k in env if k in _PROFILER_ENV or k.startswith(_PROFILER_ENV_PREFIXES)]:
        env.pop(k, None)  # (never hand a tool library to the child's launcher chain)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        files = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
        if r.returncode != 0 or not files:
            print(f"[bench] rocprofv3 child failed (rc {r.returncode}): {r.stderr[-400:]}", file=sys.stderr)
            return None
        rows = sorted(csv.DictReader(open(files[0])), key=lambda q: int(q["Start_Timestamp"]))
        steps, cur = [], None
        for q in rows:
            n = _short_kernel_name(q["Kernel_Name"])
            if first_kernel in n and (cur is None or len(cur) > 4):
                cur = []
                steps.append(cur)
            if cur is not None:
                cur.append((n, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) * 1e-9))
        if len(steps) < 8:
            return None
        lens = sorted(len(st) for st in steps)
        L = lens[len(lens) // 2]
        steps = [st for st in steps if len(st) == L][-20:]
        res = []
        for i in range(L):
            d = sorted(st[i][1] for st in steps)
            res.append((steps[-1][i][0], d[len(d) // 2]))
        return res
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rocprofv3 child: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_roofline(step, args, phases=False, graph=False):
    """Instrumented pass with the library's built-in kernel timer: every kernel launch of libspcl_hip.so is bracketed
    by HIP events ON ITS LAUNCH STREAM (spcl_profile_* of the C ABI; eager launches) and reported with its kernel symbol --
    the names rocprofv3 prints -- and the algorithmic bytes / FLOPs its entry point declares.

    ``graph=True`` (the timed configuration, N = 1): the DURATIONS are then replaced, launch by launch, by those of the same
    launches inside the replayed hipGraph (``graph_kernel_times``: a child run of this command under rocprofv3
    --kernel-trace, the method of tools/step_timeline.py) -- the configuration `value` was measured in and the committed
    profiles/rNN_kernel_stats_bench_graph.csv shows.  Eager durations (N > 1 on rank 0, --no-graph, no profiler) run ~5-10 %
    above them.  Returns the roofline of the kernel symbol with the largest total time per step, a breakdown, the step's
    kernel time."""
    from spcl_amd import native
    reps = 5
    epocher = getattr(step, "epocher", None)
    step()
    torch.cuda.synchronize()
    native.call("spcl_profile_enable", 1)
    marks, orig_fwd = [], None
    if phases and epocher is not None:  # launch-log index where the encoder forward of each instrumented step ends
        orig_fwd = epocher._forward_pass

        def marked_forward(**kw):
            start = native.call("spcl_profile_count")
            out = orig_fwd(**kw)
            marks.append((start, native.call("spcl_profile_count")))
            return out
        epocher._forward_pass = marked_forward
    source = "eager launches (HIP events on the launch stream)"
    try:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        launches = _read_profile_log(0, native.call("spcl_profile_count"))
        reps_read = reps
    finally:
        native.call("spcl_profile_enable", 0)
        if orig_fwd is not None:
            del epocher._forward_pass  # (the instance attribute shadowing the method)
    if graph and len(launches) % reps == 0:
        per = len(launches) // reps
        one = launches[-per:]                       # the last instrumented step: names, bytes, FLOPs in launch order
        gt = graph_kernel_times(args)
        if gt is not None:
            lib = [(n, t) for n, t in gt if n.startswith("spcl::")]
            names_e = [_short_kernel_name(l[0]) for l in one]
            # the eager entry points launch exactly the kernels the capture recorded, in the same order (a one-thread tick
            # or a memset node of the graph that is not a library launch is skipped by the name walk)
            j, matched, hit = 0, [], 0
            for i, ne in enumerate(names_e):
                k = j
                while k < len(lib) and k < j + 4 and lib[k][0] != ne:
                    k += 1
                if k < len(lib) and k < j + 4:  # found within the next few graph launches: its in-graph duration
                    matched.append((one[i][0], lib[k][1], one[i][2], one[i][3]))
                    j, hit = k + 1, hit + 1
                else:  # (the eager pass launches spcl_flip_pair where the product step launches spcl_flip_pair_stage)
                    matched.append(one[i])
            if hit >= 0.9 * per:
                first = len(launches) - per
                marks = [(a - first, b - first) for a, b in marks if a >= first]
                launches, reps_read = matched, 1
                source = (f"inside the replayed hipGraph for {hit} of {per} launches: rocprofv3 --kernel-trace of a child run "
                          "of this command, median of 20 replays per launch (tools/step_timeline.py's method); the rest eager")
            else:
                print(f"[bench] graph trace did not line up with the eager launch log ({hit} of {per}); eager "
                      "durations kept", file=sys.stderr)
    reps = reps_read
    peak_tf = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    groups = {}
    for sym, t, by, fl in launches:
        g = groups.setdefault(sym, {"t": 0.0, "n": 0, "bytes": 0.0, "flops": 0.0, "roof": 0.0, "hbm_t": 0.0, "mfma_t": 0.0})
        hbm_t, mfma_t = by / (HBM_PEAK_GBS * 1e9), fl / (peak_tf * 1e12)
        g["t"] += t
        g["n"] += 1
        g["bytes"] += by
        g["flops"] += fl
        g["roof"] += max(hbm_t, mfma_t)
        g["hbm_t"] += hbm_t
        g["mfma_t"] += mfma_t
    total = sum(g["t"] for g in groups.values()) / reps
    ranked = sorted(groups.items(), key=lambda kv: -kv[1]["t"])
    pmc, pmc_src = _pmc_table()

    def describe(sym, g):
        avg = g["t"] / g["n"]
        traffic = pmc.get(sym, {}).get("hbm_bytes_per_launch_corrected")
        if g["hbm_t"] >= g["mfma_t"]:
            ach = g["bytes"] / g["t"] / 1e9
            d = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic}
        else:
            ach = g["flops"] / g["t"] / 1e12
            d = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s",
                 "frac": round(ach / peak_tf, 4), "traffic": traffic}
        abytes = g["bytes"] / g["n"]
        d.update({"kernel": sym, "avg_us": round(avg * 1e6, 2), "launches_per_step": g["n"] / reps,
                  "us_per_step": round(g["t"] / reps * 1e6, 1),
                  "algorithmic_bytes_per_launch": int(abytes),
                  "algorithmic_flops_per_launch": float(g["flops"] / g["n"]),
                  "GBps": round(g["bytes"] / g["t"] / 1e9, 0), "mfma_tflops": round(g["flops"] / g["t"] / 1e12, 1),
                  "mixed_roofline_frac": round(g["roof"] / g["t"], 4),
                  "pmc_traffic_over_algorithmic": (round(traffic / abytes, 3) if traffic and abytes > 0 else None),
                  "share_of_instrumented_step": round(g["t"] / reps / total, 4)})
        return d
    out, top3 = None, []
    for sym, g in ranked:
        if g["bytes"] <= 0:
            continue  # kernels without a declared cost (latency-bound glue) cannot carry a roofline
        d = describe(sym, g)
        if out is None:
            out = dict(d)
            out.update({"traffic_source": (pmc_src + " (committed PMC profile, not this run)"
                                           if d["traffic"] is not None else None),
                        "timing_source": source,
                        "note": "kernel symbol with the largest total time among the library's launches in one step (the "
                                "ranking rocprofv3 --stats makes), HIP events on the launch stream via spcl_profile_*; bytes "
                                "/ FLOPs are the algorithmic figures of DESIGN.md section 3 summed over its launches; frac "
                                "is against the roofline that bounds it (bound), mixed_roofline_frac against max(bytes / "
                                "8 TB/s, FLOPs / peak) per launch"})
        top3.append(d)
        if len(top3) == 3:
            break
    breakdown = [{"kernel": k, "us_per_step": round(g["t"] / reps * 1e6, 1), "launches": g["n"] / reps,
                  "GBps": round(g["bytes"] / g["t"] / 1e9, 0) if g["bytes"] > 0 else None,
                  "TFLOPs": round(g["flops"] / g["t"] / 1e12, 1) if g["flops"] > 0 else None} for k, g in ranked[:16]]
    small = [l for l in launches if l[1] < 8e-6]
    tax = {"launches_per_step": len(launches) / reps, "launches_under_8us": len(small) / reps,
           "their_us_per_step": round(sum(l[1] for l in small) / reps * 1e6, 1), "timing_source": source}
    if not phases:
        return out, breakdown, total
    fr = phase_fractions(launches, marks, reps, args)
    fr["top3"] = top3
    fr["launch_tax"] = tax
    # ---- by kernel FAMILY (all template instantiations of one __global__ together): the ranking by symbol splits
    # conv3x3_fast_kernel<...> into its ~17 per-layer instantiations, none of which can be "dominant", although together they
    # own ~45 % of the step.  frac = sum over the family's launches of max(bytes / 8 TB/s, FLOPs / peak)
    # / their summed time: each launch against the roofline that bounds IT.
    fams = {}
    for sym, g in groups.items():
        f = fams.setdefault(sym.split("<")[0], {"t": 0.0, "n": 0, "roof": 0.0, "hbm_t": 0.0, "mfma_t": 0.0, "bytes": 0.0,
                                                 "flops": 0.0, "members": 0, "traffic": 0.0, "traffic_known": True})
        for k in ("t", "n", "roof", "hbm_t", "mfma_t", "bytes", "flops"):
            f[k] += g[k]
        f["members"] += 1
        tr = pmc.get(sym, {}).get("hbm_bytes_per_launch_corrected")
        if tr is None:
            f["traffic_known"] = False
        else:
            f["traffic"] += tr * g["n"]
    fam_rank = sorted(((k, f) for k, f in fams.items() if f["roof"] > 0), key=lambda kv: -kv[1]["t"])

    def fam_row(k, f):
        hbm = f["hbm_t"] >= f["mfma_t"]  # the bound that owns more of the family's summed bound time
        ach = f["bytes"] / f["t"] / 1e9 if hbm else f["flops"] / f["t"] / 1e12
        peak = HBM_PEAK_GBS if hbm else peak_tf
        return {"kernel": k + "<...>", "family": k, "instantiations": f["members"], "launches_per_step": f["n"] / reps,
                "bound": "hbm" if hbm else "mfma", "achieved": round(ach, 1 if hbm else 2), "peak": peak,
                "unit": "GB/s" if hbm else "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": (int(f["traffic"] / f["n"]) if f["traffic_known"] and f["n"] else None),
                "avg_us": round(f["t"] / f["n"] * 1e6, 2),
                "algorithmic_bytes_per_launch": int(f["bytes"] / f["n"]),
                "algorithmic_flops_per_launch": float(f["flops"] / f["n"]),
                "us_per_step": round(f["t"] / reps * 1e6, 1), "share_of_instrumented_step": round(f["t"] / reps / total, 4),
                "bound_us_per_step": round(f["roof"] / reps * 1e6, 1), "mixed_roofline_frac": round(f["roof"] / f["t"], 4),
                "GBps": round(f["bytes"] / f["t"] / 1e9, 0), "mfma_tflops": round(f["flops"] / f["t"] / 1e12, 1),
                "pmc_traffic_over_algorithmic": (round(f["traffic"] / f["bytes"], 3)
                                                 if f["traffic_known"] and f["bytes"] > 0 else None)}
    if fam_rank:
        top = fam_row(*fam_rank[0])
        top.update({"others": [fam_row(k, f) for k, f in fam_rank[1:6]], "timing_source": source,
                    "traffic_source": (pmc_src + " (committed PMC profile, not this run)" if top["traffic"] is not None else None),
                    "note": "the kernel (one __global__ template, all its instantiations' launches together) with the largest "
                            "share of the step; achieved = summed algorithmic bytes (or FLOPs) / summed launch time, per-launch "
                            "figures are averages over its launches; bound = whichever of the two rooflines owns more of the "
                            "launches' summed bound time; mixed_roofline_frac = sum over launches of max(bytes / 8 TB/s, "
                            "FLOPs / peak) / summed time -- each launch against the roofline that bounds IT"})
        fr["roofline_family"] = top
    return out, breakdown, total, fr


def phase_fractions(launches, marks, reps, args):
    """SURVEY 8(d)'s fractions from the same instrumented pass: the encoder forward (launches between the marks) against
    its mixed roofline (sum over launches of max(bytes / 8 TB/s, FLOPs / peak): 90.8 us at N = 64, 224^2, bf16) and
    against the matrix peak, the four matrix-bound layers one by one, and the whole step's mixed bound (the caller
    divides it by the timed step)."""
    peak = (MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS) * 1e12
    bw = HBM_PEAK_GBS * 1e9
    roof = lambda by, fl: max(by / bw, fl / peak)  # noqa: E731
    step_roof = sum(roof(by, fl) for _, _, by, fl in launches) / reps
    if not marks:
        return {"_step_roof_s": step_roof}
    t_fwd = sum(sum(l[1] for l in launches[a:b]) for a, b in marks) / len(marks)
    fl_fwd = sum(sum(l[3] for l in launches[a:b]) for a, b in marks) / len(marks)
    roof_fwd = sum(sum(roof(l[2], l[3]) for l in launches[a:b]) for a, b in marks) / len(marks)
    n_fwd = sum(b - a for a, b in marks) / len(marks)
    per_layer = {}
    names = ["C1a", "C1b", "C2a", "C2b", "C3a", "C3b", "C4a", "C4b", "C5a", "C5b"]
    for a, b in marks:
        convs = [l for l in launches[a:b] if l[3] > 0 and "conv3x3" in l[0]]  # (the image autocorrelation declares FLOPs too)
        if len(convs) != len(names):
            per_layer = None
            break
        for nm, l in zip(names, convs):
            d = per_layer.setdefault(nm, {"t": 0.0, "fl": 0.0, "by": 0.0})
            d["t"] += l[1]
            d["fl"] += l[3]
            d["by"] += l[2]
    layers, survey_us = None, None
    if per_layer:
        # SURVEY 8(d)'s bound counts the ten convolutions only (each reads its input once and writes its output once, the
        # BatchNorm / ReLU / pooling work fused away): 90.8 us at N = 64, 224^2, bf16
        survey_us = sum(roof(d["by"], d["fl"]) for d in per_layer.values()) / len(marks) * 1e6
    if per_layer:
        layers = {nm: {"us": round(d["t"] / len(marks) * 1e6, 2), "TFLOPs": round(d["fl"] / d["t"] / 1e12, 1),
                       "frac_of_mfma_peak": round(d["fl"] / d["t"] / peak, 4),
                       "GBps": round(d["by"] / d["t"] / 1e9, 0)} for nm, d in per_layer.items()}
    return {"_step_roof_s": step_roof,
            "encoder_fwd": {"t_us": round(t_fwd * 1e6, 1), "launches": n_fwd,
                            "mixed_roofline_us": round(roof_fwd * 1e6, 1), "mixed_frac": round(roof_fwd / t_fwd, 4),
                            "survey_bound_us": None if survey_us is None else round(survey_us, 1),
                            "survey_mixed_frac": None if survey_us is None else round(survey_us * 1e-6 / t_fwd, 4),
                            "mfma_util": round(fl_fwd / t_fwd / peak, 4), "gflop": round(fl_fwd / 1e9, 1),
                            "per_layer": layers,
                            "note": "sum of the kernel durations between the start of UNet.forward and its return (the "
                                    "instrumented pass of `roofline`, HIP events per launch: see its timing_source); mixed_frac = sum of max(bytes / 8 TB/s, "
                                    "FLOPs / peak) over ALL those launches (BatchNorm passes included) / that time; "
                                    "survey_mixed_frac = the same bound over the ten convolutions only (SURVEY 8d: 90.8 us "
                                    "at N=64 224^2 bf16) / that time"}}


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(args):
    """The oracle's restatement of the same step on the host cores: bounded samples at bs=8 (BASELINE configs[0]'s CPU
    shape) and at the GPU leg's bs, same gamma / labels / optimizer settings as the GPU leg."""
    from oracle import spcl_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 32)  # torch's CPU conv stops scaling (and collapses) far below 256 threads
    torch.set_num_threads(threads)
    gamma = 3.0 + 67.0 * (40 / 80) ** 0.5  # the GPU leg's mid-schedule age parameter (build_step)

    def sample(bs, budget_s):
        g = torch.Generator().manual_seed(1234)
        sd = O.init_unet_state(1, 4, 256, seed=10, encoder_only=True)
        sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
              for k, v in sd.items()}
        psd = {k: v.requires_grad_(True) for k, v in O.init_projector_state(256, 256, 256, seed=11).items()}
        leaves = [v for v in list(sd.values()) + list(psd.values()) if v.is_floating_point() and v.requires_grad]
        opt = torch.optim.RAdam(leaves, lr=5e-7 * 400, weight_decay=1e-5)
        img = torch.rand(bs, 1, args.size, args.size, generator=g)
        img_tf = torch.rand(bs, 1, args.size, args.size, generator=g)
        labels = [i % 3 for i in range(bs)]

        def one():
            r = O.pretrain_step(img, img_tf, sd, psd, labels, gamma=gamma, mode="soft", correct_grad=True)
            for k, p in list(sd.items()) + list(psd.items()):
                if k in r["grads"] and r["grads"][k] is not None:
                    p.grad = r["grads"][k]
            opt.step()
            opt.zero_grad()

        one()  # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < budget_s and n < 40):
            one()
            n += 1
        dt = time.perf_counter() - t0
        return bs * n / dt, n, dt

    small = sample(args.cpu_bs, 8.0)
    out = {"value": round(small[0], 2), "unit": "slices/s", "cores": threads, "kind": "port",
           "host_cpu_count": cores,
           "sample": f"{small[1]} steps of the oracle's pre-train step (fwd+bwd+RAdam) at bs={args.cpu_bs} "
                     f"({2 * args.cpu_bs} images {args.size}x{args.size}), fp32, gamma={gamma:.1f} soft, torch CPU "
                     f"{threads} threads of {cores} host CPUs, {small[2]:.1f}s",
           "parallel_info": " | ".join(ln.strip() for ln in torch.__config__.parallel_info().splitlines()
                                       if "threads" in ln or "OpenMP" in ln or "MKL" in ln)[:400]}
    if args.bs != args.cpu_bs:
        big = sample(args.bs, 8.0)
        out["at_gpu_batch"] = {"value": round(big[0], 2), "bs": args.bs,
                               "sample": f"{big[1]} steps at bs={args.bs}, {big[2]:.1f}s"}
    return out


# ------------------------------------------------------------------------------------------------ main
def spawn_ranks(args):
    """``python bench.py --gpus N`` (N > 1) without a launcher: start ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <same flags>`` as a CHILD process -- before this process makes any GPU call (it never
    does: it only waits), never an exec -- forward the ranks' output (rank 0 prints the one JSON line) and leave with the
    child's exit code.  The reference's own seam for this is ``mp.spawn(main_worker, nprocs=ngpus_per_node, ...)``
    (semi_seg/main_infonce.py:35,39)."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)  # own process group: launcher + ranks end together

    def _forward(sig, _frame):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, _forward)
    limit = float(os.environ.get("SPCL_BENCH_SPAWN_LIMIT_S", "1500"))
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"[bench] the {args.gpus}-rank child did not finish within {limit:.0f} s; killing its process group",
              file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        proc.wait()
        rc = 3
    sys.exit(rc if rc >= 0 else 128 - rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # a line that says n_gpus = 1 for a run asked to measure N would be a wrong measurement
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}, or plain `python bench.py --gpus {args.gpus}` (it spawns its ranks)",
                  file=sys.stderr, flush=True)
        sys.exit(2)  # (before any GPU call and before the process group exists)
    wd = _NoWatchdog()
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a stalled peer must end the run, not hang it: bounded process-group timeout + the heartbeat watchdog
        limit = float(os.environ.get("SPCL_BENCH_WATCHDOG_S", "60"))
        pg_timeout = datetime.timedelta(seconds=max(2 * limit, 120.0))
        # SPCL_BENCH_ONE_DEVICE=1 (self-test on a one-GPU box): every rank on cuda:0 over gloo, to exercise the N>1 control
        # flow (broadcast, flat bucket all-reduce, capture fallbacks, rank-0 reporting) -- not a measurement
        one_device = os.environ.get("SPCL_BENCH_ONE_DEVICE") == "1"
        if one_device:
            local = 0
        torch.cuda.set_device(local)
        if one_device:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
        wd = Watchdog(limit)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local if world > 1 else 0)

    if args.workload == "contrastive":
        return bench_contrastive(args, device)
    if args.workload == "finetune":
        return bench_finetune(args, device)

    if args.workload == "prostate":  # BASELINE.json configs[3] shape unless overridden on the command line
        if "--bs" not in sys.argv:
            args.bs = 64
        if "--size" not in sys.argv:
            args.size = 256
    wd.beat("build")
    step, epocher, nparams = build_step(args, device, rank, world)
    run = step
    wd.beat("capture")
    # the step captures ITSELF (epocher.step: two eager steps, then capture + replay).  Those steps run here, ahead of
    # the W warm-up steps, so that the timed region holds replays only whatever W is.
    capture_steps = 0
    if not args.no_graph:
        while capture_steps < 4 and not (epocher._step_graph is not None and epocher._step_graph.captured):
            run()
            capture_steps += 1
            wd.beat()
        sg = epocher._step_graph
        if sg is None or not sg.captured:
            if rank == 0:
                print("[bench] the epocher did not capture its step (see the warning above); running eagerly",
                      file=sys.stderr)
    sg = epocher._step_graph
    used_graph = False if (sg is None or not sg.captured) else \
        (("epocher-split" if len(sg._graphs) == 2 else f"epocher-split{len(sg._graphs)}") if world > 1 else "epocher")
    ddp_check = None
    if world > 1 and os.environ.get("SPCL_BENCH_DDP_CHECK") == "1":
        wd.beat("ddp check")
        ddp_check = check_ddp_mean(step, world, device)
    wd.beat("warmup")
    for _ in range(args.warmup):
        run()
        wd.beat()
    torch.cuda.synchronize()
    wd.beat("timed")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    gc_log = []
    if os.environ.get("SPCL_BENCH_HOSTDIAG") == "1":  # which host stalls sit in the timed region: garbage collections?
        import gc
        _gc_t = [0.0]

        def _gc_cb(phase, info):
            if phase == "start":
                _gc_t[0] = time.perf_counter()
            else:
                gc_log.append((len(host), info.get("generation"), round((time.perf_counter() - _gc_t[0]) * 1e3, 2)))
        gc.callbacks.append(_gc_cb)
    t0 = time.perf_counter()
    host = []
    for _ in range(args.steps):
        h0 = time.perf_counter()
        run()
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    if os.environ.get("SPCL_BENCH_HOSTDIAG") == "1" and rank == 0:
        slow = sorted(((round(h * 1e3, 2), i) for i, h in enumerate(host)), reverse=True)[:8]
        print(f"[bench hostdiag] slowest host steps (ms, index): {slow}; gc (step, generation, ms): {gc_log}", file=sys.stderr)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    wd.beat("reduce")
    multi = None
    if world > 1:
        # every rank's own clock (a slow rank or a slow link shows here) ...
        per_rank = torch.zeros(world, device=device, dtype=torch.float64)
        per_rank[rank] = elapsed / args.steps * 1e3
        dist.all_reduce(per_rank)
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # ... and the step's one collective alone: HIP events around the epocher's exchange (flat bucket, communication
        # stream, both stream waits included), median of 20 after 5 warm calls, MAX over the ranks
        xs = []
        for k in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            epocher.step_exchange()
            e1.record()
            e1.synchronize()
            if k >= 5:
                xs.append(e0.elapsed_time(e1) * 1e3)
        xs.sort()
        ar = torch.tensor([xs[len(xs) // 2]], device=device, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        multi = {"per_rank_ms_per_step": [round(float(v), 4) for v in per_rank.tolist()],
                 "allreduce_us": round(float(ar.item()), 1),
                 "allreduce_note": "the flat gradient bucket's all-reduce alone on an idle GPU, HIP events on the caller's stream "
                                   "(median of 20, max over ranks)"}
    # per-replay distribution AFTER the timed region (HIP events on the launch stream around every single replay):
    # median / p10 / p90 of one step's GPU time -- an extra key, `value` stays the wall-clock aggregate above
    replay = None
    if not args.no_extras:
        wd.beat("replay distribution")
        replay = replay_distribution(run, max(args.steps, 50) if world == 1 else args.steps, wd)
    ms = elapsed / args.steps * 1e3
    value = args.bs * world * args.steps / elapsed
    prostate = args.workload == "prostate"
    line = {
        "metric": ("pretrain slices/sec/node (UNet + 3 combined InfoNCE hooks, 256^2, bs=64/GPU)" if prostate else
                   "pretrain slices/sec/node (UNet+InfoNCE, 224^2, bs=32/GPU)"), "value": round(value, 1),
        "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[3] shape (SURVEY row N4): one encoder pass to Conv5, three "
                                "self-paced hooks (partition / patient / self meta-labels, own projector each) summed, "
                                "fwd+bwd + RAdam" if prostate else
                                "BASELINE.json configs[1]: self-paced pretrain step, UNet base (max_channel=256) encoder "
                                "to Conv5 + ProjectionHead(256,256,256) + SelfPacedSupConLoss(soft, partition labels, "
                                "correct_grad) fwd+bwd + RAdam") + (" + flat RCCL grad all-reduce" if world > 1 else ""),
                   "slices_per_gpu": args.bs, "images_per_gpu_step": 2 * args.bs, "image": f"1x{args.size}x{args.size}",
                   "global_batch": args.bs * world, "parallelism": f"dp{world}", "params": nparams,
                   "hipgraph": used_graph, "capture_steps": capture_steps,
                   "batches": ("drawn on device every step (torch.rand)" if args.fresh_rand else
                               f"{args.pool} distinct resident batches cycled") + ", fresh flip seed and label vector "
                              "every step (PretrainEncoderEpocher.step on next(loader))"},
    }
    if rank == 0:
        loss = epocher.meters.statistics()
        line["final_meters"] = {k: round(v["mean"], 5) for g in loss.values() for k, v in g.items()
                                if k in ("loss", "sp_weight", "reg_loss")}
    hs = sorted(host)
    line["host_us_per_step"] = {"median": round(hs[len(hs) // 2] * 1e6, 1), "max": round(hs[-1] * 1e6, 1),
                                "note": "host time to enqueue one step (stage refill + upload, flip launch, graph replay); "
                                        "it must stay below the GPU's step time for the queue to run ahead"}
    if replay is not None:
        line["replay_us"] = replay
    if ddp_check is not None:
        line["ddp_check"] = ddp_check
    if multi is not None:
        line["multi_gpu"] = multi
    if not args.no_roofline and rank == 0:
        # rank 0 alone runs the instrumented steps, WITHOUT the collective (it is not a kernel of this library): compute
        # + update phases only, so no peer is needed and a failure here cannot leave another rank inside a collective
        wd.beat("roofline")
        try:
            if args.ddp_overlap:  # the instrumented steps run on this rank alone: no early collective from backward
                from spcl_amd import ddp as _ddp
                _ddp.disable_unet_overlap(step.flat, step.model)
            in_graph = world == 1 and bool(used_graph) and os.environ.get("SPCL_BENCH_GRAPH_TRACE", "1") != "0"
            # in_graph: the durations are those of the launches inside a replayed step (see measure_roofline)
            roof, breakdown, tot, fractions = measure_roofline(local_step(step), args, phases=True, graph=in_graph)
            line["roofline"] = roof
            line["kernel_breakdown"] = breakdown
            line["instrumented_step_ms"] = round(tot * 1e3, 3)
            if fractions:
                fractions["step_mixed_frac"] = round(fractions.pop("_step_roof_s") / (ms * 1e-3), 4)
                enc = fractions.get("encoder_fwd") or {}
                if enc.get("survey_bound_us"):
                    # the ten convolutions' sum of max(bytes / 8 TB/s, FLOPs / peak) forward (90.8 us at N = 64,
                    # 224^2, bf16), x 3 for forward + input gradient + weight gradient, against the TIMED step
                    fractions["step_survey_frac"] = round(3.0 * enc["survey_bound_us"] * 1e-3 / ms, 4)
                    fractions["step_survey_bound_us"] = round(3.0 * enc["survey_bound_us"], 1)
                if "roofline_family" in fractions:
                    # the HEADLINE roofline is the kernel that owns the largest share of the step with all its template
                    # instantiations together (conv3x3_fast_kernel: 17 launches, 42 %); ranked by symbol those 15
                    # instantiations can never be "dominant" and the line would show a stream kernel that is 8 % of the step
                    #  -- that one stays beside it as `roofline_symbol`
                    line["roofline_symbol"] = roof
                    line["roofline"] = fractions.pop("roofline_family")
                line.setdefault("extra", {}).update(fractions)
        except Exception as e:  # noqa: BLE001
            line["roofline"] = None
            print(f"[bench] roofline pass failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
    wd.beat("final barrier")
    if world > 1:
        dist.barrier()
    wd.stop()
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            line.setdefault("extra", {})["contrastive_4096x128"] = contrastive_numbers(device, steps=50, warmup=10)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] contrastive extra failed: {type(e).__name__}: {e}", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_extras and args.dtype == "bf16" and args.workload == "pretrain":
        try:  # the reference's own storage type end to end (fp32 tensors, split-bf16 products): a short pass of the same step
            line.setdefault("extra", {})["fp32_step"] = fp32_numbers(args, device)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] fp32 extra failed: {type(e).__name__}: {e}", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "pretrain":
        # the two other workloads the framework is benchmarked on (tools/profile_round_all.sh writes their full lines), as short
        # graphed passes so that the driver's line carries them too
        for key, fn in (("finetune_step", finetune_numbers), ("prostate_step", prostate_numbers)):
            try:
                line.setdefault("extra", {})[key] = fn(args, device)
            except Exception as e:  # noqa: BLE001
                print(f"[bench] {key} extra failed: {type(e).__name__}: {e}", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def _f32_split():
    import spcl_amd  # noqa: F401
    from spcl_amd import native
    return native.call("spcl_conv_get_f32_split")


def fp32_numbers(args, device, steps=12, warmup=4):
    """`--dtype fp32` in short: the same pre-train step with fp32 storage (the parity mode: torch's default float32, as the
    reference runs), through the same epocher and its hipGraph, fresh batches per step.  The convolutions multiply every f32
    operand as three bf16 pieces (six bf16 MFMAs per product, f32-grade: spcl_conv_set_f32_split, include/spcl_hip.h); the
    exact-f32 MFMA (1/16 of the bf16 matrix rate) stays one call away."""
    import copy
    a = copy.copy(args)
    a.dtype = "fp32"
    step, epocher, _ = build_step(a, device, 0, 1)
    for _ in range(4):
        step()
        if epocher._step_graph is not None and epocher._step_graph.captured:
            break
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sg = epocher._step_graph
    out = {"ms_per_step": round(dt * 1e3, 4), "slices_s": round(a.bs / dt, 1), "steps": steps,
           "hipgraph": bool(sg is not None and sg.captured), "dtype": "fp32",
           "f32_split": int(_f32_split()),
           "note": "same workload, fp32 storage; convolutions multiply f32 operands as three bf16 pieces (six "
                   "v_mfma_f32_16x16x32_bf16 per product, dropped terms < 2^-24: f32-grade, tests hold the f32 tolerances); "
                   "f32_split 0 = v_mfma_f32_16x16x4_f32 (exact f32 products, 1/16 of the bf16 matrix rate)"}
    from spcl_amd import stepgraph as _sg
    _sg.gc_release()
    del step, epocher
    torch.cuda.empty_cache()
    return out


def _short_pass(step, graph_of, steps, warmup):
    """``steps`` timed calls of ``step`` after the epocher has captured itself and ``warmup`` replays -> seconds per step"""
    for _ in range(4):
        step()
        sg = graph_of()
        if sg is not None and sg.captured:
            break
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def prostate_numbers(args, device, steps=20, warmup=5):
    """`--workload prostate` in short (BASELINE.json configs[3] shape: 256^2, bs 64, three combined self-paced hooks)."""
    import copy
    a = copy.copy(args)
    a.workload, a.bs, a.size = "prostate", 64, 256
    step, epocher, _ = build_step(a, device, 0, 1)
    dt = _short_pass(step, lambda: epocher._step_graph, steps, warmup)
    sg = epocher._step_graph
    out = {"ms_per_step": round(dt * 1e3, 4), "slices_s": round(a.bs / dt, 1), "steps": steps, "dtype": a.dtype,
           "hipgraph": bool(sg is not None and sg.captured),
           "workload": "BASELINE.json configs[3] shape: 64 x 2 images of 256^2, three self-paced hooks on one encoder pass"}
    from spcl_amd import stepgraph as _sg
    _sg.gc_release()
    del step, epocher
    torch.cuda.empty_cache()
    return out


def finetune_numbers(args, device, steps=20, warmup=5):
    """`--workload finetune` in short (SURVEY N1 / BASELINE.json configs[2]'s fine-tune half: full UNet, 32 x 224^2)."""
    import copy
    a = copy.copy(args)
    a.workload, a.bs, a.size, a.no_graph = "finetune", 32, 224, False
    step, ep = build_finetune_step(a, device)
    dt = _short_pass(step, lambda: ep._step_graph, steps, warmup)
    sg = ep._step_graph
    out = {"ms_per_step": round(dt * 1e3, 4), "slices_s": round(a.bs / dt, 1), "steps": steps, "dtype": a.dtype,
           "hipgraph": bool(sg is not None and sg.captured),
           "workload": "fine-tune step: full UNet fwd+bwd, softmax + KL_div on one-hot labels, Dice counts, RAdam; 32 x 224^2"}
    from spcl_amd import stepgraph as _sg
    _sg.gc_release()
    del step, ep
    torch.cuda.empty_cache()
    return out


def check_ddp_mean(step, world, device):
    """N > 1 self-test (SPCL_BENCH_DDP_CHECK=1): every rank differentiates ITS batch, the flat gradients of all ranks are
    gathered, and the step's one collective must leave exactly their mean in every rank's bucket (a sum of `world`
    terms in rank order, divided by `world`; bit for bit at two ranks).  No optimizer step."""
    epocher, flat = step.epocher, step.flat
    early_idx = flat._early_idx  # two-bucket overlap: the rank's OWN gradient is taken from a pass without the early collective
    flat._early_idx = None
    with epocher.meters.focus_on(epocher.meter_focus):
        epocher.step_compute(step.batch, seed=7)
    local = flat.flat.clone()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    if early_idx is not None:  # ... and the same batch again (the step is deterministic), the early bucket sent from backward
        flat._early_idx = early_idx
        with epocher.meters.focus_on(epocher.meter_focus):
            epocher.step_compute(step.batch, seed=7)
    epocher.step_exchange()
    mean = parts[0].clone()
    for p in parts[1:]:
        mean += p
    mean /= world
    # (with fold_mean the bucket holds the ranks' SUM and the optimizer kernel applies grad_scale = 1 / world)
    diff = float((flat.flat * flat.grad_scale - mean).abs().max())
    differ = float((parts[0] - parts[-1]).abs().max())  # the ranks really saw different batches
    from spcl_amd.contrastyou import meters as _meters
    _meters.flush_batch()
    return {"max_abs_diff_vs_mean_of_rank_gradients": diff, "max_abs_diff_between_ranks": differ,
            "grad_abs_max": float(mean.abs().max())}


def local_step(step):
    """the step without its collective (compute + update phases of the epocher): what one GPU's kernels do"""
    epocher, batch = step.epocher, step.batch

    def run():
        from spcl_amd import stepgraph as _sg

        def body():
            with epocher.meters.focus_on(epocher.meter_focus):
                epocher.step_update(epocher.step_compute(batch, seed=7))
        if next(epocher._model.parameters()).is_cuda:  # (on the stream every other backward pass of this model ran on)
            _sg.run_on_side_stream(body)
        else:
            body()
    run.epocher = epocher
    return run


def replay_distribution(run, k, wd=None):
    """GPU time of single steps: HIP events (torch's current stream == the launch stream of every kernel and of the
    graph replay) around each of ``k`` further steps."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        run()
        b.record()
        if wd is not None:
            wd.beat()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    pick = lambda q: round(us[min(len(us) - 1, int(q * len(us)))], 1)  # noqa: E731
    return {"n": k, "median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9), "min": round(us[0], 1),
            "note": "HIP-event time of single steps, measured after the timed region"}


def build_finetune_step(args, device):
    """-> (step, epocher): one iteration of FineTuneEpocher's loop on the synthetic labelled loader's next batch"""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import FineTuneEpocher
    from spcl_amd.synthetic import SyntheticLabeledLoader
    torch.manual_seed(10)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = UNet(input_dim=1, num_classes=4, max_channel=256).to(device)
    model.set_compute_dtype(dtype)
    flat = ddp.FlatParams([p for p in model.parameters() if p.requires_grad])
    opt = FusedRAdam([flat.param], lr=1e-4, weight_decay=1e-5)
    loader = SyntheticLabeledLoader(bs=args.bs, size=args.size, device=device, seed=77, pool=args.pool)
    ep = FineTuneEpocher(model=model, optimizer=opt, labeled_loader=loader, sup_criterion=KL_div(), num_batches=10 ** 9,
                         device=device, flat_params=flat, graph=not args.no_graph)
    model.train()

    def step():
        """one iteration of the epocher's loop on the loader's NEXT batch (own images and label maps, all resident)"""
        with ep.meters.focus_on(ep.meter_focus):
            return ep.step(next(loader))

    return step, ep


def bench_finetune(args, device):
    """SURVEY row N1 (BASELINE.json configs[2], the fine-tune half): full UNet fwd+bwd + softmax/KL_div + RAdam on
    synthetic labelled 224^2 slices, single GPU, hipGraph.  One slice = one image here (no second view)."""
    step, ep = build_finetune_step(args, device)
    run = step  # FineTuneEpocher.step captures itself after two eager steps (stepgraph.py)
    for _ in range(0 if args.no_graph else 3):
        run()
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    stats = ep.meters.statistics()["semi"]
    line = {"metric": "fine-tune slices/sec (full UNet + KL_div, 224^2)", "value": round(args.bs / dt, 1),
            "unit": "slices/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "SURVEY N1 / BASELINE.json configs[2] fine-tune half: UNet base (max_channel=256) full "
                                   "forward+backward, softmax + KL_div on one-hot labels, RAdam",
                       "slices_per_gpu": args.bs, "image": f"1x{args.size}x{args.size}",
                       "batches": f"{args.pool} distinct resident labelled batches cycled (FineTuneEpocher.step on next(loader))",
                       "hipgraph": "epocher" if (ep._step_graph is not None and ep._step_graph.captured) else False},
            "final_meters": {"sup_loss": round(stats["sup_loss"]["mean"], 5),
                             "sup_dice": {k: round(v, 4) for k, v in stats["sup_dice"].items()}}}
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline_finetune(args)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] fine-tune CPU baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
    if not args.no_roofline:
        try:
            ep._graph_on = False  # the instrumented pass launches every kernel eagerly
            roof, breakdown, tot = measure_roofline(step, args)
            line["roofline"], line["kernel_breakdown"] = roof, breakdown
            line["instrumented_step_ms"] = round(tot * 1e3, 3)
        except Exception as e:  # noqa: BLE001
            line["roofline"] = None
            print(f"[bench] roofline pass failed: {type(e).__name__}: {e}", file=sys.stderr)
    print(json.dumps(line))


def cpu_baseline_finetune(args, budget_s=8.0):
    """The oracle's restatement of the fine-tune step (full UNet forward + softmax / KL_div + backward + RAdam,
    new_epocher.py:260-283) on the host cores: a bounded sample at bs = --cpu-bs."""
    from oracle import spcl_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    bs = args.cpu_bs
    g = torch.Generator().manual_seed(77)
    sd = O.init_unet_state(1, 4, 256, seed=10)
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    leaves = [v for v in sd.values() if v.is_floating_point() and v.requires_grad]
    opt = torch.optim.RAdam(leaves, lr=1e-4, weight_decay=1e-5)
    img = torch.rand(bs, 1, args.size, args.size, generator=g)
    lab = torch.randint(0, 4, (bs, args.size, args.size), generator=g)

    def one():
        loss = O.finetune_loss(O.unet_forward(img, sd, train=True), lab)
        opt.zero_grad()
        loss.backward()
        opt.step()

    one()
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 40):
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 2), "unit": "slices/s", "cores": threads, "kind": "port", "host_cpu_count": cores,
            "sample": f"{n} steps of the oracle's fine-tune step (full UNet fwd+bwd + KL_div + RAdam) at bs={bs} "
                      f"({args.size}x{args.size}), fp32, torch CPU {threads} threads of {cores} host CPUs, {dt:.1f}s"}


def contrastive_numbers(device, steps=50, warmup=10):
    """BASELINE.json configs[4]: SelfPacedSupConLoss (soft, gamma 12, correct_grad) at 2n=4096 samples, proj_dim=128,
    forward and forward+backward, eager and replayed from a hipGraph (HIP events on the launch stream)."""
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
    n, d = 2048, 128
    g = torch.Generator().manual_seed(1)
    z1 = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(device).requires_grad_(True)
    z2 = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(device).requires_grad_(True)
    labels = (torch.arange(n, device=device) % 3).float()  # device float labels, as the InfoNCE hook hands them over
    crit = SelfPacedSupConLoss(weight_update="soft", correct_grad=True, sync_checks=False)
    crit.set_gamma(12.0)

    def fwd():
        return crit(z1, z2, target=labels)

    def fwdbwd():
        loss = fwd()
        loss.backward()

    def timed(fn):
        for _ in range(warmup):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return us[len(us) // 2]  # median, us

    res, eager, REP = {}, {}, 10
    for name, fn in (("fwd", fwd), ("fwd_bwd", fwdbwd)):
        eager[name] = timed(fn)  # one launch after the other from Python: host-bound once the kernels are short
        # the same work replayed from a hipGraph (as the training step is): GPU time
        z1.grad = z2.grad = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
                z1.grad = z2.grad = None
        torch.cuda.current_stream().wait_stream(side)
        # REP passes per graph: a replay's own start-up (the GPU idles ~8 us while the graph's first packet is
        # processed) is not part of the loss, which runs inside the training step's graph
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(REP):
                fn()
                z1.grad = z2.grad = None
        res[name] = timed(graph.replay) / REP
    mat_bytes = 3 * 4096 * 4096 * 4 + 2 * 4096 * 128 * 4
    ach = mat_bytes / (res["fwd"] * 1e-6) / 1e9
    return {"fwd_us": round(res["fwd"], 1), "fwd_bwd_us": round(res["fwd_bwd"], 1),
            "eager_fwd_us": round(eager["fwd"], 1), "eager_fwd_bwd_us": round(eager["fwd_bwd"], 1),
            "frac": round(ach / HBM_PEAK_GBS, 4), "achieved_GBps": round(ach, 1), "algorithmic_bytes": mat_bytes,
            "note": "median replay of a hipGraph holding 10 passes, / 10 (HIP events); frac = SURVEY 8(d)'s "
                    "materialised-fp32 schedule bytes (205.5 MB) / forward time / 8 TB/s -- the forward itself is fused "
                    "(two split-bf16 MFMA sweeps, no logits matrix), so its own HBM traffic is a few MB"}


def bench_contrastive(args, device):
    """BASELINE.json configs[4] microbench: loss only, 2n=4096 samples, proj_dim=128."""
    r = contrastive_numbers(device, steps=args.steps, warmup=args.warmup)
    line = {"metric": "contrastive similarity+softmax 4096x128 (microbench)", "value": round(1e6 / r["fwd_bwd_us"], 1),
            "unit": "loss fwd+bwd /s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(r["fwd_bwd_us"] * 1e-3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[4]: SelfPacedSupConLoss 2n=4096 d=128", "hipgraph": True,
                       "fwd_us": r["fwd_us"], "fwd_bwd_us": r["fwd_bwd_us"],
                       "eager_fwd_us": r["eager_fwd_us"], "eager_fwd_bwd_us": r["eager_fwd_bwd_us"]},
            "roofline": {"bound": "hbm", "achieved": r["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": r["frac"], "traffic": None,
                         "note": "SURVEY 8(d) accounting: the materialised-fp32 schedule's bytes (205.5 MB: S written "
                                 "once, read for the row sums, read for the weighted NLL) / forward time"}}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
