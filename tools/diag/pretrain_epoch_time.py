"""Wall clock of pre-train EPOCHS in the reference's own configuration (config/pretrain.yaml: 200 batches per epoch; the
trainer builds a new epocher every epoch and the self-paced hook moves its age parameter): what PretrainEncoderTrainer does
per epoch (trainers/pretrain.py _create_tra_epoch), on the synthetic loader, bs 32 at 224^2."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402,F401
from spcl_amd import ddp  # noqa: E402
from spcl_amd.optim import FusedRAdam  # noqa: E402
from spcl_amd.semi_seg.arch import UNet  # noqa: E402
from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher  # noqa: E402
from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks  # noqa: E402
from spcl_amd.synthetic import SyntheticPretrainLoader  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(10)
model = UNet(input_dim=1, num_classes=4, max_channel=256, momentum=0.1).to(dev)
model.set_compute_dtype(torch.bfloat16)
hook = create_sp_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition", begin_values=3.0,
                               end_values=70.0, mode="soft", max_epoch=80, p=0.5, correct_grad=True, data_name="acdc",
                               sync_checks=False).to(dev)
for sub in hook._hooks:
    sub._scheduler.epoch = 40
for name in model.decoder_names:
    getattr(model, "_" + name).requires_grad_(False)
params = [p for p in model.parameters() if p.requires_grad] + list(hook.parameters())
flat = ddp.FlatParams(params)
opt = FusedRAdam([flat.param], lr=5e-7 * 400, weight_decay=1e-5)
loader = SyntheticPretrainLoader(bs=32, size=224, device=dev, seed=1234, resident=True, meta="acdc", pool=8)
if len(sys.argv) > 2 and sys.argv[2] == "real":
    # the product's own data path: a device-resident store of 8-bit slices, the reference's contrastive batch
    # sampler (config/pretrain.yaml: 10 scans x 3 partitions = 30 slices), both views augmented on the device every step
    from spcl_amd.semi_seg.data import get_contrastive_dataloader, synthetic_slice_store
    store = synthetic_slice_store(device="cuda", scans=100, slices_per_scan=(9, 12), size=256, seed=6)
    loader = iter(get_contrastive_dataloader(store, {"scan_sample_num": 10, "partition_sample_num": 1})[0])
model.train()
for epoch in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ep = PretrainEncoderEpocher(model=model, optimizer=opt, chain_dataloader=loader, num_batches=200, device=dev,
                                inference_until="Conv5", flat_params=flat, cur_epoch=epoch)
    ep.add_hooks([hook()])  # (the trainer-level hook hands out the epoch's hook: the age parameter follows its schedule)
    ts, tl = [], []
    with ep.meters.focus_on(ep.meter_focus):
        for i in range(200):
            a = time.perf_counter()
            batch = next(loader)
            tl.append(time.perf_counter() - a)
            ep.step(batch)
            if i < 4:
                torch.cuda.synchronize()
            ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    ep.close_hooks()
    tot = time.perf_counter() - t0
    print(f"epoch {epoch}: total {tot * 1e3:.1f} ms; steps 0-3 (synchronised) {[round(t * 1e3, 2) for t in ts[:4]]} ms; "
          f"(total - first four) / 196 = {(tot - sum(ts[:4])) / 196 * 1e3:.3f} ms; host per step: loader "
          f"{sorted(tl[4:])[98] * 1e6:.0f} us, loader + step {sorted(ts[4:])[98] * 1e6:.0f} us", flush=True)
