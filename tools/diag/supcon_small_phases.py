"""where the one-workgroup loss kernel (2n = 64) spends its time: SPCL_SUPCON_DBG early exits (16 after the load phase, 32
after the S tiles, 64 after the forward scalars), kernel durations from the library's own timer"""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
if len(sys.argv) == 1:
    for dbg in (128, 256, 16, 32, 64, 0):
        for d in (64, 256):
            env = dict(os.environ, SPCL_SUPCON_DBG=str(dbg))
            subprocess.run([sys.executable, __file__, str(d)], env=env)
    sys.exit(0)
import torch
import spcl_amd  # noqa
from spcl_amd import native
from spcl_amd import functional as F
d = int(sys.argv[1])
RAW = os.environ.get("PHASES_RAW", "0") == "1"  # the rows before F.normalize (spcl_supcon_forward_rows)
n = 32
g = torch.Generator().manual_seed(1)
z = torch.nn.functional.normalize(torch.randn(2 * n, d, generator=g), dim=1).cuda()
labels = (torch.arange(n) % 3).float().cuda()
def run():
    st = F.SupConState()
    F.supcon_loss(z, None, labels, None, t=0.07, sp_mode=F.SP_SOFT, gamma=12.0, correct_grad=True, state=st,
                  normalize_inputs=RAW)
for _ in range(5):
    run()
torch.cuda.synchronize()
native.call("spcl_profile_enable", 1)
for _ in range(20):
    run()
torch.cuda.synchronize()
cnt = native.call("spcl_profile_count")
name = ctypes.create_string_buffer(256)
us, by, fl = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
ts = []
for i in range(cnt):
    native.call("spcl_profile_get", i, name, 256, ctypes.byref(us), ctypes.byref(by), ctypes.byref(fl))
    if b"supcon_small" in name.value:
        ts.append(us.value)
native.call("spcl_profile_enable", 0)
ts.sort()
print(f"raw={int(RAW)} dbg={os.environ.get('SPCL_SUPCON_DBG')} d={d}: supcon_small median {ts[len(ts)//2]:.1f} us (min {ts[0]:.1f})")
