"""Diagnostic: per-layer relative L2 error of the bf16 HIP encoder against the bf16-emulating oracle and fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spcl_amd
from spcl_amd.semi_seg.arch import UNet
from oracle import spcl_oracle as O

torch.set_num_threads(os.cpu_count())
for (shape, mc) in [((2, 1, 224, 224), 256), ((4, 1, 56, 56), 256)]:
    sd = O.init_unet_state(shape[1], 4, mc, seed=21)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(22))
    print("shape", shape)
    for until in ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5"):
        m = UNet(input_dim=shape[1], num_classes=4, max_channel=mc)
        m.load_state_dict(sd)
        m.cuda().train().set_compute_dtype(torch.bfloat16)
        with torch.no_grad():
            y = m(x.cuda(), until=until).float().cpu()
            ye = O.unet_forward(x, {k: v.clone() for k, v in sd.items()}, until, q=O.BF16Emulation)
            yf = O.unet_forward(x, {k: v.clone() for k, v in sd.items()}, until)
        nz = (y != ye).float().mean()
        print(f"  {until}: vs emu relL2 {float((y-ye).norm()/ye.norm()):.5f} (frac differing {float(nz):.4f}); "
              f"vs fp32 {float((y-yf).norm()/yf.norm()):.5f}; emu vs fp32 {float((ye-yf).norm()/yf.norm()):.5f}")
