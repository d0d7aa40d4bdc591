"""GPU parity of the HIP projector (avg-pool -> MLP -> L2 normalise, fwd+bwd) vs golden vectors and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spcl_oracle as O


def _head(ci, ch, co, seed, head_type="mlp", normalize=True):
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead
    h = ProjectionHead(input_dim=ci, hidden_dim=ch, output_dim=co, head_type=head_type, normalize=normalize)
    sd = O.init_projector_state(ci, ch, co, seed=seed, head_type=head_type)
    h.load_state_dict(sd, strict=True)
    return h.cuda(), sd


@pytest.mark.parametrize("tag", ["small", "base"])
@pytest.mark.parametrize("layout", ["nchw", "channels_last", "bf16_cl"])
def test_projector_golden(golden, tag, layout):
    g = golden("g2_projector.npz")
    ci, ch, co, seed = [int(v) for v in g[f"{tag}/dims"]]
    head, _ = _head(ci, ch, co, seed)
    x = torch.tensor(g[f"{tag}/x"]).cuda()
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    if layout == "bf16_cl":
        x = x.contiguous(memory_format=torch.channels_last).bfloat16()
    x.requires_grad_(True)
    y = head(x)
    (y * torch.tensor(g[f"{tag}/r"]).cuda()).sum().backward()
    tol = dict(rtol=1e-4, atol=1e-5) if layout != "bf16_cl" else dict(rtol=3e-2, atol=3e-3)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}/y"], **tol)
    gtol = dict(rtol=1e-3, atol=1e-6) if layout != "bf16_cl" else dict(rtol=5e-2, atol=2e-4)
    assert x.grad.shape == x.shape
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), g[f"{tag}/dx"], **gtol)
    for k, p in head.named_parameters():
        ref = g[f"{tag}/grad/{k}"]
        a = 1e-4 * float(np.abs(ref).max()) if layout != "bf16_cl" else 3e-2 * float(np.abs(ref).max())
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3 if layout != "bf16_cl" else 5e-2, atol=a,
                                   err_msg=k)


@pytest.mark.parametrize("head_type,normalize", [("linear", True), ("mlp", False)])
def test_projector_variants_vs_oracle(head_type, normalize):
    head, sd = _head(24, 40, 12, 9, head_type, normalize)
    x = torch.randn(5, 24, 7, 7, generator=torch.Generator().manual_seed(4)).relu()
    r = torch.randn(5, 12, generator=torch.Generator().manual_seed(5))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.projector_forward(xr, sdg, head_type=head_type, normalize=normalize)
    (yr * r).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = head(xg)
    (y * r.cuda()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-6)
    for k, p in head.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), sdg[k].grad.numpy(), rtol=1e-3, atol=1e-5, err_msg=k)


def test_projector_scope_errors():
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead
    # adaptive_max exists since round 2 (heads.py:26-45); pooling to anything but (1, 1) cannot feed the flat MLP head:
    # the reference fails in its first Linear, here the forward says why
    head = ProjectionHead(input_dim=8, output_dim=8, head_type="mlp", normalize=True, pool_name="adaptive_max").cuda()
    assert head(torch.rand(2, 8, 4, 4).cuda()).shape == (2, 8)
    bad = ProjectionHead(input_dim=8, output_dim=8, head_type="mlp", normalize=True, pool_name="adaptive_avg",
                         spatial_size=(2, 2)).cuda()
    with pytest.raises(RuntimeError):
        bad(torch.rand(2, 8, 4, 4).cuda())
    with pytest.raises(AssertionError):
        ProjectionHead(input_dim=8, output_dim=8, head_type="conv", normalize=True)


@pytest.mark.parametrize("K,dtype", [(2, torch.float32), (3, torch.bfloat16), (4, torch.float32)])
def test_batched_heads_equal_the_single_head_calls(K, dtype):
    """functional.projector_heads (one launch per layer for K heads on one feature) against K functional.projector calls:
    identical z per head, identical parameter gradients, the feature gradient = the sum of the K single-head ones."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_hip
    g = torch.Generator().manual_seed(K)
    N, C, H, W, hid, out = 10, 64, 6, 5, 48, 24
    feat = torch.randn(N, C, H, W, generator=g).cuda().to(dtype)
    heads = [[(torch.randn(hid, C, generator=g) / 8).cuda(), (torch.randn(hid, generator=g) / 8).cuda(),
              (torch.randn(out, hid, generator=g) / 7).cuda(), (torch.randn(out, generator=g) / 7).cuda()]
             for _ in range(K)]
    up = [torch.randn(N, out, generator=g).cuda() for _ in range(K)]

    def run(batched):
        f = feat.clone().requires_grad_(True)
        hs = [[t.clone().requires_grad_(True) for t in h] for h in heads]
        zs = F_hip.projector_heads(f, hs) if batched else [F_hip.projector(f, *h) for h in hs]
        sum((z * u).sum() for z, u in zip(zs, up)).backward()
        return [z.detach() for z in zs], f.grad, [[t.grad for t in h] for h in hs]

    z0, gf0, gp0 = run(False)
    z1, gf1, gp1 = run(True)
    for a, b in zip(z0, z1):
        assert torch.equal(a, b)
    for ha, hb in zip(gp0, gp1):
        for a, b in zip(ha, hb):
            assert torch.equal(a, b)
    tol = 1e-5 if dtype == torch.float32 else 2e-2  # the K gradients are added in f32 before the one bf16 rounding
    assert float((gf0.float() - gf1.float()).abs().max()) <= tol * float(gf0.float().abs().max())


def test_projector_call_time_normalize_false_returns_the_rows_before_normalisation():
    """``head(x, normalize=False)`` (the hook's form in front of ``criterion(..., normalize_inputs=True)``): the rows whose
    F.normalize the ordinary call returns, bit for bit, and the same parameter / input gradients through
    ``l2norm_rows`` as through the head's own normalisation"""
    import spcl_amd.functional as F_hip
    head, _ = _head(64, 48, 32, 3)
    x = torch.randn(6, 64, 5, 5, generator=torch.Generator().manual_seed(1)).relu().cuda()
    r = torch.randn(6, 32, generator=torch.Generator().manual_seed(2)).cuda()
    xa = x.clone().requires_grad_(True)
    za = head(xa)
    (za * r).sum().backward()
    ga = {k: p.grad.clone() for k, p in head.named_parameters()}
    head.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_(True)
    ob = head(xb, normalize=False)
    zb = F_hip.l2norm_rows(ob)
    (zb * r).sum().backward()
    assert torch.equal(za, zb)
    assert not torch.equal(ob, zb)
    assert torch.equal(xa.grad, xb.grad)
    for k, p in head.named_parameters():
        assert torch.equal(p.grad, ga[k]), k
