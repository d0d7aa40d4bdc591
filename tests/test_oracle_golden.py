"""The oracle (oracle/spcl_oracle.py) is pinned here against outputs of the reference itself
(tests/golden/*.npz, written by tools/gen_golden.py from the imported reference) and against the
analytic known-answer relations KAT-1..5 of SURVEY section 4.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import spcl_oracle as O

MODES = {"supcon1": (None, None, False), "hard_1e6": ("hard", 1e6, False), "hard_7": ("hard", 7.0, False),
         "soft_12": ("soft", 12.0, False), "soft_12_cg": ("soft", 12.0, True), "soft_3_cg": ("soft", 3.0, True)}


def labels_of(name, n):
    return {"mod3": [i % 3 for i in range(n)], "distinct": list(range(n)),
            "acdc": sorted(i % 3 for i in range(n)), "none": None}[name]


def parse_case(key):
    parts = key.split("_")
    n, d = int(parts[0][1:]), int(parts[1][1:])
    lname = parts[2]
    mname = "_".join(parts[3:])
    return n, d, lname, mname


def test_g1_loss_matches_reference(golden):
    g = golden("g1_loss.npz")
    for key in g["cases"]:
        key = str(key)
        n, d, lname, mname = parse_case(key)
        mode, gamma, cg = MODES[mname]
        z1 = torch.tensor(g[f"n{n}_d{d}/z1"], requires_grad=True)
        z2 = torch.tensor(g[f"n{n}_d{d}/z2"], requires_grad=True)
        r = O.supcon_loss(z1, z2, labels_of(lname, n), gamma=gamma, mode=mode or "hard", correct_grad=cg)
        r["loss"].backward()
        np.testing.assert_allclose(r["loss"].item(), g[f"{key}/loss"], rtol=2e-6, atol=1e-7, err_msg=key)
        np.testing.assert_allclose(z1.grad.numpy(), g[f"{key}/dz1"], rtol=1e-4, atol=2e-6, err_msg=key)
        np.testing.assert_allclose(z2.grad.numpy(), g[f"{key}/dz2"], rtol=1e-4, atol=2e-6, err_msg=key)
        if mode is not None:
            np.testing.assert_allclose(float(r["rho"]), g[f"{key}/rho"], rtol=1e-6, err_msg=key)
        if n <= 8:
            np.testing.assert_allclose(r["sim_logits"].detach().numpy(), g[f"{key}/sim_logits"], atol=1e-5)
            np.testing.assert_allclose(r["sim_exp"].detach().numpy(), g[f"{key}/sim_exp"], rtol=1e-5, atol=1e-9)
            np.testing.assert_array_equal(r["pos_mask"].numpy(), g[f"{key}/pos_mask"])
            if mode is not None:
                np.testing.assert_allclose(r["sp_mask"].numpy(), g[f"{key}/sp_mask"], atol=1e-6)
        # closed-form gradient (fp64) agrees with the reference's autograd gradient
        a, b = O.supcon_grad(z1.detach().double(), z2.detach().double(), labels_of(lname, n), gamma=gamma,
                             mode=mode or "hard", correct_grad=cg)
        if mname != "hard_7":  # hard thresholds can flip between fp32/fp64 at the boundary
            np.testing.assert_allclose(a.numpy(), g[f"{key}/dz1"], rtol=2e-3, atol=5e-6, err_msg=key)


def test_g8_wide_projection_loss_matches_reference(golden):
    """the oracle at projection widths beyond 256 (fixture written from the reference, tools/gen_golden.py wide)"""
    g = golden("g8_wide.npz")
    for key in g["wide/cases"]:
        key = str(key)
        n, d, lname, mname = parse_case(key.split("/", 1)[1])
        mode, gamma, cg = MODES[mname]
        z1 = torch.tensor(g[f"wide/n{n}_d{d}/z1"], requires_grad=True)
        z2 = torch.tensor(g[f"wide/n{n}_d{d}/z2"], requires_grad=True)
        r = O.supcon_loss(z1, z2, labels_of(lname, n), gamma=gamma, mode=mode or "hard", correct_grad=cg)
        r["loss"].backward()
        np.testing.assert_allclose(r["loss"].item(), g[f"{key}/loss"], rtol=2e-6, atol=1e-7, err_msg=key)
        np.testing.assert_allclose(z1.grad.numpy(), g[f"{key}/dz1"], rtol=1e-4, atol=2e-6, err_msg=key)
        np.testing.assert_allclose(z2.grad.numpy(), g[f"{key}/dz2"], rtol=1e-4, atol=2e-6, err_msg=key)
        if mode is not None:
            np.testing.assert_allclose(float(r["rho"]), g[f"{key}/rho"], rtol=1e-6, err_msg=key)


def test_g1_mask_input(golden):
    g = golden("g1_loss.npz")
    z1 = torch.tensor(g["mask_n6_d32/z1"], requires_grad=True)
    z2 = torch.tensor(g["mask_n6_d32/z2"], requires_grad=True)
    r = O.supcon_loss(z1, z2, mask=torch.tensor(g["mask_n6_d32/mask"]), gamma=9.0, mode="soft", correct_grad=True)
    r["loss"].backward()
    np.testing.assert_allclose(r["loss"].item(), g["mask_n6_d32/loss"], rtol=2e-6)
    np.testing.assert_allclose(float(r["rho"]), g["mask_n6_d32/rho"], rtol=1e-6)
    np.testing.assert_allclose(z1.grad.numpy(), g["mask_n6_d32/dz1"], rtol=1e-4, atol=2e-6)


def test_g2_projector(golden):
    g = golden("g2_projector.npz")
    for tag in ("small", "base"):
        ci, ch, co, seed = [int(v) for v in g[f"{tag}/dims"]]
        sd = O.init_projector_state(ci, ch, co, seed=seed)
        assert math.isclose(sum(float(v.double().sum()) for v in sd.values()), float(g[f"{tag}/param_checksum"]),
                            rel_tol=1e-12)
        if tag == "small":
            for k, v in sd.items():
                np.testing.assert_array_equal(v.numpy(), g[f"{tag}/param/{k}"])
        sd = {k: v.requires_grad_(True) for k, v in sd.items()}
        x = torch.tensor(g[f"{tag}/x"], requires_grad=True)
        y = O.projector_forward(x, sd)
        (y * torch.tensor(g[f"{tag}/r"])).sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/y"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}/dx"], rtol=1e-4, atol=1e-7)
        for k, v in sd.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"{tag}/grad/{k}"], rtol=1e-4, atol=1e-6)


def test_g3_encoder(golden):
    g = golden("g3_encoder.npz")
    sd0 = O.init_unet_state(1, 4, 128, seed=11)
    assert math.isclose(sum(float(v.double().sum()) for v in sd0.values()), float(g["small/param_checksum"]),
                        rel_tol=1e-12)
    x = torch.tensor(g["small/x"])
    for until in ("Conv1", "Conv2", "Conv3", "Conv4"):
        sd = {k: v.clone() for k, v in sd0.items()}
        y = O.unet_forward(x, sd, until)
        np.testing.assert_allclose(y.numpy(), g[f"small/out/{until}"], rtol=1e-4, atol=1e-5)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in sd0.items()}
    y = O.encoder_forward(x, sd, "Conv5")
    np.testing.assert_allclose(y.detach().numpy(), g["small/out/Conv5"], rtol=1e-4, atol=1e-5)
    (y * torch.tensor(g["small/r"])).sum().backward()
    for k in g.files:
        if k.startswith("small/grad/"):
            name = k[len("small/grad/"):]
            ref = g[k]
            tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(sd[name].grad.numpy(), ref, rtol=1e-3, atol=tol, err_msg=name)
        if k.startswith("small/buf/"):
            name = k[len("small/buf/"):]
            np.testing.assert_allclose(sd[name].detach().numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=name)
    # eval mode with the updated stats
    with torch.no_grad():
        ye = O.encoder_forward(x, sd, "Conv5", train=False)
    np.testing.assert_allclose(ye.numpy(), g["small/eval_out/Conv5"], rtol=1e-4, atol=1e-5)
    # full network (row N1)
    sdf = {k: v.clone() for k, v in sd0.items()}
    np.testing.assert_allclose(O.unet_forward(x, sdf).numpy(), g["small/out/full"], rtol=1e-4, atol=1e-5)
    with pytest.raises(KeyError):
        O.unet_forward(x, sdf, "Conv9")


def test_g3_encoder_base_checksum(golden):
    g = golden("g3_encoder.npz")
    sd = O.init_unet_state(1, 4, 256, seed=21)
    assert math.isclose(sum(float(v.double().sum()) for v in sd.values()), float(g["base/param_checksum"]),
                        rel_tol=1e-12)
    x = torch.rand(2, 1, 224, 224, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        y = O.encoder_forward(x, sd, "Conv5")
    np.testing.assert_allclose(y.mean(dim=(0, 2, 3)).numpy(), g["base/out_mean_c"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y[0].numpy(), g["base/out_n0"], rtol=1e-3, atol=1e-4)


def test_g4_full_step(golden):
    g = golden("g4_step.npz")
    cmax, hid, od, s1, s2 = [int(v) for v in g["dims"]]
    sd = O.init_unet_state(1, 4, cmax, seed=s1)
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k and k.startswith("_Conv")
              else v) for k, v in sd.items()}
    psd = {k: v.requires_grad_(True) for k, v in O.init_projector_state(cmax, hid, od, seed=s2).items()}
    r = O.pretrain_step(torch.tensor(g["img"]), torch.tensor(g["img_tf"]), sd, psd, g["labels"].tolist(),
                        gamma=10.0, mode="soft", correct_grad=True)
    np.testing.assert_allclose(r["loss"].item(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(float(r["rho"]), g["rho"], rtol=1e-5)
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            mine = r["grads"][name[5:] if name.startswith("proj.") else name]
            ref = g[k]
            tol = 2e-4 * max(1e-6, float(np.abs(ref).max()))
            np.testing.assert_allclose(mine.numpy(), ref, rtol=2e-3, atol=tol, err_msg=name)


# ---------------------------------------------------------------------------- KATs
def _unit(n, d, seed, dtype=torch.float32):
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(seed), dtype=dtype)
    return z / z.norm(dim=1, keepdim=True)


def test_kat1_selfpaced_hard_huge_gamma_is_supcon1():
    for (n, d) in [(8, 256), (30, 256)]:
        z1, z2 = _unit(n, d, 1), _unit(n, d, 2)
        lab = [i % 3 for i in range(n)]
        a = O.supcon_loss(z1, z2, lab, gamma=1e6, mode="hard")["loss"]
        b = O.supcon_loss(z1, z2, lab)["loss"]
        assert torch.equal(a, b)


def test_kat2_orthonormal_simclr_closed_form():
    n, d = 8, 32
    q, _ = torch.linalg.qr(torch.randn(d, d, generator=torch.Generator().manual_seed(3), dtype=torch.float64))
    P = q[:2 * n]
    loss = O.supcon_loss(P[:n], P[n:], None)["loss"]
    assert abs(loss.item() - math.log(2 * n - 1)) < 1e-9


def test_kat3_soft_limits():
    z1, z2 = _unit(8, 64, 4), _unit(8, 64, 5)
    lab = [i % 2 for i in range(8)]
    r0 = O.supcon_loss(z1, z2, lab, gamma=1e-9, mode="soft", correct_grad=True)
    assert r0["loss"].item() == 0 and float(r0["rho"]) == 0
    rinf = O.supcon_loss(z1, z2, lab, gamma=1e12, mode="soft")["loss"]
    assert abs(rinf.item() - O.supcon_loss(z1, z2, lab)["loss"].item()) < 1e-5


def test_kat4_gradient_finite_difference_fp64():
    n, d = 5, 12
    z1, z2 = _unit(n, d, 6, torch.float64), _unit(n, d, 7, torch.float64)
    lab = [0, 1, 0, 1, 2]
    # (a) w == 1: the closed form is the true derivative -> central differences agree
    g1, g2 = O.supcon_grad(z1, z2, lab)
    eps = 1e-6
    for (i, k) in [(0, 0), (2, 5), (4, 11)]:
        zp, zm = z1.clone(), z1.clone()
        zp[i, k] += eps
        zm[i, k] -= eps
        fd = (O.supcon_loss(zp, z2, lab)["loss"] - O.supcon_loss(zm, z2, lab)["loss"]) / (2 * eps)
        assert abs(fd.item() - g1[i, k].item()) < 1e-7
        zp, zm = z2.clone(), z2.clone()
        zp[i, k] += eps
        zm[i, k] -= eps
        fd = (O.supcon_loss(z1, zp, lab)["loss"] - O.supcon_loss(z1, zm, lab)["loss"]) / (2 * eps)
        assert abs(fd.item() - g2[i, k].item()) < 1e-7
    # (b) self-paced weights are constants of the step (no-grad): closed form == autograd
    for kw in (dict(gamma=6.0, mode="soft", correct_grad=False), dict(gamma=6.0, mode="soft", correct_grad=True),
               dict(gamma=2.5, mode="hard", correct_grad=True)):
        g1, g2 = O.supcon_grad(z1, z2, lab, **kw)
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        O.supcon_loss(a, b, lab, **kw)["loss"].backward()
        assert (a.grad - g1).abs().max().item() < 1e-12 and (b.grad - g2).abs().max().item() < 1e-12


def test_kat5_bn_train_statistics():
    sd = O.init_unet_state(1, 4, 128, seed=5, encoder_only=True)
    x = torch.rand(4, 1, 16, 16, generator=torch.Generator().manual_seed(6))
    w = sd["_Conv1.conv.0.weight"]
    y = torch.nn.functional.conv2d(x, w, None, 1, 1)
    O.unet_forward(x, sd, "Conv1")
    m = y.mean(dim=(0, 2, 3))
    v_unbiased = y.var(dim=(0, 2, 3), unbiased=True)
    np.testing.assert_allclose(sd["_Conv1.conv.1.running_mean"].numpy(), 0.1 * m.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd["_Conv1.conv.1.running_var"].numpy(), 0.9 + 0.1 * v_unbiased.numpy(), rtol=1e-5)
    assert int(sd["_Conv1.conv.1.num_batches_tracked"]) == 1


def test_pscheduler_and_labels():
    s = O.PScheduler(80, 3, 70, 0.5)
    vals = []
    for e in range(80):
        if e in (0, 1, 40, 79):
            vals.append(s.value)
        s.step()
    np.testing.assert_allclose(vals, [3.0, 3 + 67 * math.sqrt(1 / 80), 3 + 67 * math.sqrt(0.5),
                                      3 + 67 * math.sqrt(79 / 80)], rtol=1e-12)
    groups = ["patient004_00", "patient004_01", "patient001_00", "patient010_01"]
    parts = ["2", "0", "1", "0"]
    assert O.get_label("partition", "acdc", parts, groups) == [2, 0, 1, 0]
    assert O.get_label("patient", "acdc", parts, groups) == [1, 1, 0, 2]
    assert O.get_label("cycle", "acdc", parts, groups) == [0, 1, 0, 1]
    assert O.get_label("self", "acdc", parts, groups) == [0, 1, 2, 3]
    assert O.get_label("patient", "prostate", parts, ["Case00_0", "Case01_3"]) == [0, 1]
    with pytest.raises(NotImplementedError):
        O.get_label("cycle", "prostate", parts, groups)


def test_g5_decoder_and_finetune_step():
    """Oracle's full UNet (decoder) + fine-tune loss vs the reference's modules (tools/gen_golden.py gen_decoder)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g5_decoder.npz"))
    sd = O.init_unet_state(1, 4, 128, seed=11)
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["param_checksum"])) < 1e-6
    x, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["labels"])
    for until in ("Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"):
        sdc = {k: v.clone() for k, v in sd.items()}
        np.testing.assert_allclose(O.unet_forward(x, sdc, until, train=True).numpy(), g[f"out/{until}"], rtol=1e-4,
                                   atol=2e-5)
    sdc = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    logits = O.unet_forward(x, sdc, None, train=True)
    np.testing.assert_allclose(logits.detach().numpy(), g["out/logits"], rtol=1e-4, atol=2e-5)
    loss = O.finetune_loss(logits, labels)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k in g.files:
        if k.startswith("grad/"):
            ref = g[k]
            got = sdc[k[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 2e-4 * max(1e-6, np.abs(ref).max()) + 1e-7, k
        elif k.startswith("buf/") and "num_batches" not in k:
            np.testing.assert_allclose(sdc[k[4:]].detach().numpy(), g[k], rtol=1e-4, atol=1e-6)
    ev = O.unet_forward(x, {k: v.detach() for k, v in sdc.items()}, None, train=False)
    np.testing.assert_allclose(ev.detach().numpy(), g["eval/logits"], rtol=1e-4, atol=2e-5)
    assert (ev.max(1)[1].numpy() == g["eval/pred"]).mean() > 0.999


def test_dice_counts_and_universal_dice_known_answers():
    pred = torch.tensor([[[0, 1], [1, 2]], [[2, 2], [0, 0]]])
    tgt = torch.tensor([[[0, 1], [2, 2]], [[2, 0], [0, 0]]])
    i, u = O.dice_counts(pred, tgt, 3)
    assert i.tolist() == [[1, 1, 1], [2, 0, 1]] and u.tolist() == [[2, 3, 3], [5, 0, 3]]
    mean, std = O.universal_dice(i, u, ["a", "a"])  # one 3-D group: counts are summed before the ratio
    np.testing.assert_allclose(mean.numpy(), [(2 * 3 + 1e-6) / (7 + 1e-6), (2 + 1e-6) / (3 + 1e-6), (4 + 1e-6) / (6 + 1e-6)],
                               rtol=1e-6)
    mean2, _ = O.universal_dice(i, u, ["a", "b"])  # slice-wise groups: an empty class scores 1 (1e-6 / 1e-6)
    np.testing.assert_allclose(float(mean2[1]), 0.5 * ((2 + 1e-6) / (3 + 1e-6) + 1.0), rtol=1e-6)


def test_round2_golden_heads_and_exclude_other_pos(golden):
    """g6 (written from the reference by tools/gen_golden.py round2): SupConLoss1(exclude_other_pos=True), the
    adaptive-max pooled ProjectionHead and DenseProjectionHead -- the oracle's restatements reproduce them."""
    g = golden("g6_round2.npz")
    for key in g["xpos/cases"]:
        key = str(key)
        n, d, lname = key.split("/")[1].split("_")[0][1:], key.split("_")[1][1:], key.split("_", 2)[2]
        n, d = int(n), int(d)
        z1 = torch.tensor(g[f"xpos/n{n}_d{d}/z1"], requires_grad=True)
        z2 = torch.tensor(g[f"xpos/n{n}_d{d}/z2"], requires_grad=True)
        loss = O.supcon_loss_exclude_other_pos(z1, z2, labels_of(lname, n))
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"{key}/loss"], rtol=2e-5, err_msg=key)
        np.testing.assert_allclose(z1.grad.numpy(), g[f"{key}/dz1"], rtol=2e-3, atol=2e-6, err_msg=key)
        np.testing.assert_allclose(z2.grad.numpy(), g[f"{key}/dz2"], rtol=2e-3, atol=2e-6, err_msg=key)
    x = torch.tensor(g["maxhead/x"], requires_grad=True)
    params = {k[len("maxhead/param/"):]: torch.tensor(g[k], requires_grad=True) for k in g.files
              if k.startswith("maxhead/param/")}
    z = O.projector_forward(x, params, pool_name="adaptive_max")
    (z * torch.tensor(g["maxhead/r"])).sum().backward()
    np.testing.assert_allclose(z.detach().numpy(), g["maxhead/z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["maxhead/dx"], rtol=1e-4, atol=1e-6)
    for tag, kw in {"dense_mlp": dict(head_type="mlp", pool_name="adaptive_avg", spatial_size=(5, 4)),
                    "dense_lin": dict(head_type="linear", pool_name="adaptive_max", spatial_size=(3, 3))}.items():
        x = torch.tensor(g[f"{tag}/x"], requires_grad=True)
        params = {k[len(tag) + 7:]: torch.tensor(g[k], requires_grad=True) for k in g.files
                  if k.startswith(f"{tag}/param/")}
        z = O.dense_projector_forward(x, params, **kw)
        (z * torch.tensor(g[f"{tag}/r"])).sum().backward()
        np.testing.assert_allclose(z.detach().numpy(), g[f"{tag}/z"], rtol=1e-5, atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}/dx"], rtol=1e-4, atol=1e-6, err_msg=tag)
        for k, p in params.items():
            np.testing.assert_allclose(p.grad.numpy(), g[f"{tag}/grad/{k}"], rtol=1e-4, atol=1e-6, err_msg=(tag, k))
