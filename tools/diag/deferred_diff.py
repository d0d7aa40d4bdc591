import sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import spcl_amd
from spcl_amd import ddp, functional as F_
from spcl_amd.contrastyou.projectors.heads import ProjectionHead
from oracle import spcl_oracle as O
from tests.test_gpu_encoder import _unet

def run(sinks, twice):
    net, _ = _unet(256, 3, torch.bfloat16)
    head = ProjectionHead(input_dim=256, hidden_dim=32, output_dim=16, head_type="mlp", normalize=True)
    head.load_state_dict(O.init_projector_state(256, 32, 16, seed=5)); head.cuda()
    for name in net.decoder_names: getattr(net, "_" + name).requires_grad_(False)
    named = [(k, p) for k, p in list(net.named_parameters()) + [("h."+k, p) for k, p in head.named_parameters()] if p.requires_grad]
    flat = ddp.FlatParams([p for _, p in named])
    x = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    z = head(net(x, until="Conv5"))
    loss = (z * torch.arange(16, device="cuda")).sum()
    if twice: loss = loss + 0.5 * head(net(x.flip(3), until="Conv5")).sum()
    if sinks: flat.zero_grad()
    loss.backward()
    flat.gather_grads()
    return {k: v.clone() for (k, _), v in zip(named, flat.views)}
for twice in (False, True):
    a, b = run(False, twice), run(True, twice)
    for k in a:
        e = float((a[k]-b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30))
        if e > 1e-5: print(twice, k, e, float(a[k].abs().max()))
