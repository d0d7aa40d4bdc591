"""Per-kernel times of the contrastive loss at 2n=4096, d=128 (library timer, eager)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spcl_amd  # noqa
from spcl_amd import native as n
from spcl_amd.contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
N, d = 2048, 128
g = torch.Generator().manual_seed(1)
z1 = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda().requires_grad_(True)
z2 = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda().requires_grad_(True)
labels = (torch.arange(N, device="cuda") % 3).float()
crit = SelfPacedSupConLoss(weight_update="soft", correct_grad=True, sync_checks=False)
crit.set_gamma(12.0)
for _ in range(3):
    crit(z1, z2, target=labels).backward()
torch.cuda.synchronize()
n.call("spcl_profile_enable", 1)
reps = 20
for _ in range(reps):
    crit(z1, z2, target=labels).backward()
torch.cuda.synchronize()
cnt = n.call("spcl_profile_count")
name = ctypes.create_string_buffer(256)
us, by, fl = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
acc, order = {}, []
for i in range(cnt):
    n.call("spcl_profile_get", i, name, 256, ctypes.byref(us), ctypes.byref(by), ctypes.byref(fl))
    k = name.value.decode()
    if k not in acc:
        order.append(k); acc[k] = [0.0, 0]
    acc[k][0] += us.value; acc[k][1] += 1
n.call("spcl_profile_enable", 0)
tot = 0
for k in order:
    t = acc[k][0] / reps
    tot += t
    print(f"{t:8.1f} us/step  x{acc[k][1] / reps:.0f}  {k[:90]}")
print(f"{tot:8.1f} us total")
