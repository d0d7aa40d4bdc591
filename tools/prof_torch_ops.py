"""Which PyTorch (non-libspcl) ops does one pre-train step launch?  Eager step under torch.profiler, ops with their
input shapes and the Python line that issued them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def build_finetune_step(dev, bs=32, size=224):
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import FineTuneEpocher
    from spcl_amd.synthetic import SyntheticLabeledLoader
    model = UNet(input_dim=1, num_classes=4, max_channel=256).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    flat = ddp.FlatParams([p for p in model.parameters() if p.requires_grad])
    opt = FusedRAdam([flat.param], lr=1e-4, weight_decay=1e-5)
    loader = SyntheticLabeledLoader(bs=bs, size=size, device=dev, seed=77)
    ep = FineTuneEpocher(model=model, optimizer=opt, labeled_loader=loader, sup_criterion=KL_div(), num_batches=10 ** 9,
                         device=dev, flat_params=flat)
    model.train()
    batch = next(loader)

    def step():
        with ep.meters.focus_on(ep.meter_focus):
            return ep.step(batch)
    return step


def main():
    args = bench.parse_args([]) if hasattr(bench, "parse_args") else None
    if args is None:
        import argparse
        args = argparse.Namespace(bs=32, size=224, dtype="bf16")
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "finetune":
        step = build_finetune_step(dev)
    else:
        step, epocher, _ = bench.build_step(args, dev, 0, 1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        t = getattr(e, "self_device_time_total", 0)
        if t <= 0 or not e.key.startswith("aten::"):
            continue
        stack = [s for s in (e.stack or []) if ("self-paced" in s or "spcl_amd" in s or "bench.py" in s)]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:70], stack[0][-100:] if stack else ""))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"aten ops with GPU self time: {len(rows)}, total {tot:.0f} us")
    for t, c, n, s, st in rows:
        print(f"{t:7.1f} x{c:<2d} {n:26s} {s:70s} {st}")


if __name__ == "__main__":
    main()
