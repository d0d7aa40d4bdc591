"""race hunt for the sweeps' LDS hand-off words: the large-batch loss and its gradient N times from the same inputs, every
result compared bit for bit with the first (2n = 4096: four tiles per workgroup; 2n = 5120 / 8192: the ring images are reused)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import spcl_amd  # noqa
from spcl_amd.contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss, SupConLoss1

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n, d, mode in ((2048, 128, "soft"), (2560, 128, "hard"), (4096, 64, "soft"), (4096, 128, None)):
    g = torch.Generator().manual_seed(n + d)
    z1 = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).cuda().requires_grad_(True)
    z2 = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).cuda().requires_grad_(True)
    labels = (torch.arange(n) % 7).float().cuda()
    crit = SupConLoss1(sync_checks=False) if mode is None else SelfPacedSupConLoss(weight_update=mode, correct_grad=True, sync_checks=False)
    if mode is not None:
        crit.set_gamma(12.0)
    first, bad = None, 0
    # a streaming kernel on a second stream keeps the memory system unevenly loaded while the sweeps run
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    for r in range(reps):
        if r % 2:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        z1.grad = z2.grad = None
        loss = crit(z1, z2, target=labels)
        loss.backward()
        cur = (loss.detach().clone(), z1.grad.clone(), z2.grad.clone())
        if first is None:
            first = cur
        elif not all(torch.equal(a, b) for a, b in zip(first, cur)):
            bad += 1
    torch.cuda.synchronize()
    print(f"2n={2 * n} d={d} mode={mode}: {reps} repeats, {bad} differ from the first; loss {float(first[0]):.6f}")
    assert bad == 0
print("ok")
