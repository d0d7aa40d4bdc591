#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation (runs only in the build container).

The reference's arithmetic modules are imported from a scratch copy of /root/reference (the import has
side effects on the tree -- SURVEY 8c pitfall) with three in-process stubs (loguru, torch._six,
deepclustering2.configparser._utils).  Only data (inputs + the reference's outputs) is written; no
reference source travels.  Inputs that would be large are regenerated from seeds by
oracle/spcl_oracle.py's init_* helpers and pinned by a checksum.

    python tools/gen_golden.py            # writes tests/golden/
"""
import collections.abc
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF_SRC = "/root/reference"
REF_TMP = "/tmp/ref"
OUT = os.path.join(REPO, "tests", "golden")


def _import_reference():
    if not os.path.isdir(REF_TMP):
        shutil.copytree(REF_SRC, REF_TMP)
    sys.path.insert(0, REF_TMP)

    def _mod(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    class _L:
        def __getattr__(s, k):
            return lambda *a, **kw: s if k in ("opt", "bind") else None

        def catch(s, *a, **kw):
            return lambda f: f

    _mod("loguru", logger=_L())
    _mod("torch._six", container_abcs=collections.abc, int_classes=int, string_classes=str)
    _mod("deepclustering2")
    _mod("deepclustering2.configparser")
    _mod("deepclustering2.configparser._utils", get_config=lambda scope="base": {})
    from contrastyou.losses.contrast_loss3 import SupConLoss1, SelfPacedSupConLoss
    from contrastyou.projectors.heads import ProjectionHead
    from semi_seg.arch.unet import UNet
    from semi_seg.arch.hook import SingleFeatureExtractor
    return SupConLoss1, SelfPacedSupConLoss, ProjectionHead, UNet, SingleFeatureExtractor


def unit_rows(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(n, d, generator=g)
    return torch.nn.functional.normalize(z, dim=1)


def label_sets(n):
    acdc = [(i % 3) for i in range(n)]  # ContrastBatchSampler: scans x 3 partitions
    return {"mod3": [i % 3 for i in range(n)], "distinct": list(range(n)),
            "acdc": sorted(acdc), "none": None}


LOSS_MODES = [  # (name, mode, gamma, correct_grad)   gamma None -> SupConLoss1
    ("supcon1", None, None, False),
    ("hard_1e6", "hard", 1e6, False),
    ("hard_7", "hard", 7.0, False),
    ("soft_12", "soft", 12.0, False),
    ("soft_12_cg", "soft", 12.0, True),
    ("soft_3_cg", "soft", 3.0, True),
]


def gen_loss(SupConLoss1, SelfPacedSupConLoss):
    out = {}
    cases = []
    for (n, d) in [(4, 16), (8, 256), (30, 256), (96, 128)]:
        z1, z2 = unit_rows(n, d, 100 + n), unit_rows(n, d, 200 + n)
        for lname, labels in label_sets(n).items():
            for (mname, mode, gamma, cg) in LOSS_MODES:
                if n >= 96 and (lname != "mod3" or mname not in ("supcon1", "hard_7", "soft_12_cg")):
                    continue  # keep the fixture small
                key = f"n{n}_d{d}_{lname}_{mname}"
                a = z1.clone().requires_grad_(True)
                b = z2.clone().requires_grad_(True)
                if mode is None:
                    crit = SupConLoss1(temperature=0.07)
                else:
                    crit = SelfPacedSupConLoss(temperature=0.07, weight_update=mode, correct_grad=cg)
                    crit.set_gamma(gamma)
                loss = crit(a, b, target=labels)
                loss.backward()
                out[f"{key}/loss"] = loss.detach().numpy()
                out[f"{key}/dz1"] = a.grad.numpy()
                out[f"{key}/dz2"] = b.grad.numpy()
                if mode is not None:
                    out[f"{key}/rho"] = np.float64(crit.downgrade_ratio)
                if n <= 8:
                    out[f"{key}/sim_logits"] = crit.sim_logits.detach().numpy()
                    out[f"{key}/sim_exp"] = crit.sim_exp.detach().numpy()
                    out[f"{key}/pos_mask"] = crit.pos_mask.numpy()
                    if mode is not None:
                        out[f"{key}/sp_mask"] = crit.sp_mask.numpy()
                cases.append(key)
        out[f"n{n}_d{d}/z1"] = z1.numpy()
        out[f"n{n}_d{d}/z2"] = z2.numpy()
    # explicit (asymmetric-free) mask case: mask given instead of target
    n, d = 6, 32
    z1, z2 = unit_rows(n, d, 7), unit_rows(n, d, 8)
    mask = (torch.arange(n)[:, None] % 2 == torch.arange(n)[None, :] % 2).float()
    a = z1.clone().requires_grad_(True)
    b = z2.clone().requires_grad_(True)
    crit = SelfPacedSupConLoss(weight_update="soft", correct_grad=True)
    crit.set_gamma(9.0)
    loss = crit(a, b, mask=mask)
    loss.backward()
    out["mask_n6_d32/z1"], out["mask_n6_d32/z2"], out["mask_n6_d32/mask"] = z1.numpy(), z2.numpy(), mask.numpy()
    out["mask_n6_d32/loss"], out["mask_n6_d32/rho"] = loss.detach().numpy(), np.float64(crit.downgrade_ratio)
    out["mask_n6_d32/dz1"], out["mask_n6_d32/dz2"] = a.grad.numpy(), b.grad.numpy()
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "g1_loss.npz"), **out)
    print("g1_loss:", len(cases), "cases")


def gen_projector(ProjectionHead):
    from oracle.spcl_oracle import init_projector_state
    out = {}
    for tag, (ci, ch, co, seed) in {"small": (16, 32, 16, 3), "base": (256, 256, 256, 4)}.items():
        head = ProjectionHead(input_dim=ci, hidden_dim=ch, output_dim=co, head_type="mlp", normalize=True)
        sd = init_projector_state(ci, ch, co, seed=seed)
        head.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(50 + ci)
        nb = 4 if tag == "small" else 2
        x = torch.randn(nb, ci, 14, 14, generator=g).relu_().requires_grad_(True)
        r = torch.randn(nb, co, generator=g)
        y = head(x)
        (y * r).sum().backward()
        out[f"{tag}/x"], out[f"{tag}/r"], out[f"{tag}/y"] = x.detach().numpy(), r.numpy(), y.detach().numpy()
        out[f"{tag}/dx"] = x.grad.numpy()
        for k, p in head.named_parameters():
            out[f"{tag}/grad/{k}"] = p.grad.numpy()
        if tag == "small":
            for k, v in sd.items():
                out[f"{tag}/param/{k}"] = v.numpy()
        out[f"{tag}/param_checksum"] = np.float64(sum(float(v.double().sum()) for v in sd.values()))
        out[f"{tag}/dims"] = np.array([ci, ch, co, seed])
    np.savez_compressed(os.path.join(OUT, "g2_projector.npz"), **out)
    print("g2_projector done")


def gen_encoder(UNet):
    from oracle.spcl_oracle import init_unet_state
    out = {}
    # small: max_channel=128 (8/16/32/64/128), [4,1,32,32]
    net = UNet(input_dim=1, num_classes=4, max_channel=128)
    sd = init_unet_state(1, 4, 128, seed=11)
    net.load_state_dict(sd, strict=True)
    net.train()
    g = torch.Generator().manual_seed(12)
    x = torch.rand(4, 1, 32, 32, generator=g)
    out["small/x"] = x.numpy()
    out["small/param_checksum"] = np.float64(sum(float(v.double().sum()) for v in sd.values()))
    for until in ("Conv1", "Conv2", "Conv3", "Conv4"):
        ref = UNet(input_dim=1, num_classes=4, max_channel=128)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        out[f"small/out/{until}"] = ref(x, until=until).detach().numpy()
    y = net(x, until="Conv5")
    out["small/out/Conv5"] = y.detach().numpy()
    r = torch.randn(y.shape, generator=g)
    out["small/r"] = r.numpy()
    (y * r).sum().backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            out[f"small/grad/{k}"] = p.grad.numpy()
    for k, b in net.named_buffers():
        if k.startswith(("_Conv",)):
            out[f"small/buf/{k}"] = b.numpy()
    # eval-mode forward with the updated running stats
    net.eval()
    out["small/eval_out/Conv5"] = net(x, until="Conv5").detach().numpy()
    # full UNet (decoder, "next" row N1) small output
    net.train()
    full = UNet(input_dim=1, num_classes=4, max_channel=128)
    full.load_state_dict(sd, strict=True)
    full.train()
    out["small/out/full"] = full(x).detach().numpy()

    # base: max_channel=256, [2,1,224,224], Conv5 checksum only
    sd2 = init_unet_state(1, 4, 256, seed=21)
    net2 = UNet(input_dim=1, num_classes=4, max_channel=256)
    net2.load_state_dict(sd2, strict=True)
    net2.train()
    x2 = torch.rand(2, 1, 224, 224, generator=torch.Generator().manual_seed(22))
    y2 = net2(x2, until="Conv5").detach()
    out["base/out_mean_c"] = y2.mean(dim=(0, 2, 3)).numpy()
    out["base/out_absmax_c"] = y2.abs().amax(dim=(0, 2, 3)).numpy()
    out["base/out_n0"] = y2[0, :, :, :].numpy()
    out["base/param_checksum"] = np.float64(sum(float(v.double().sum()) for v in sd2.values()))
    np.savez_compressed(os.path.join(OUT, "g3_encoder.npz"), **out)
    print("g3_encoder done")


def gen_step(UNet, ProjectionHead, SelfPacedSupConLoss, SingleFeatureExtractor):
    """G4: one pre-train step through the reference modules incl. the forward-hook tap (arch/hook.py)."""
    from oracle.spcl_oracle import init_unet_state, init_projector_state
    out = {}
    sd = init_unet_state(1, 4, 128, seed=31)
    psd = init_projector_state(128, 64, 32, seed=32)
    net = UNet(input_dim=1, num_classes=4, max_channel=128)
    net.load_state_dict(sd, strict=True)
    net.train()
    head = ProjectionHead(input_dim=128, hidden_dim=64, output_dim=32, head_type="mlp", normalize=True)
    head.load_state_dict(psd, strict=True)
    crit = SelfPacedSupConLoss(weight_update="soft", correct_grad=True)
    crit.set_gamma(10.0)
    g = torch.Generator().manual_seed(33)
    n = 6
    img, img_tf = torch.rand(n, 1, 32, 32, generator=g), torch.rand(n, 1, 32, 32, generator=g)
    labels = [i % 3 for i in range(n)]
    ext = SingleFeatureExtractor(net, "Conv5")
    ext.bind()
    ext.clear()
    ext.set_enable(True)
    net(torch.cat([img, img_tf], 0), until="Conv5")
    ext.set_enable(False)
    feat = ext.feature()[-2 * n:]
    z = head(feat)
    a, b = torch.chunk(z, 2)
    loss = crit(a, b, target=labels)
    loss.backward()
    ext.remove()
    out["img"], out["img_tf"], out["labels"] = img.numpy(), img_tf.numpy(), np.array(labels)
    out["loss"], out["rho"] = loss.detach().numpy(), np.float64(crit.downgrade_ratio)
    out["z"] = z.detach().numpy()
    for k, p in list(net.named_parameters()) + [("proj." + k, p) for k, p in head.named_parameters()]:
        if p.grad is not None:
            out[f"grad/{k}"] = p.grad.numpy()
    out["dims"] = np.array([128, 64, 32, 31, 32])
    np.savez_compressed(os.path.join(OUT, "g4_step.npz"), **out)
    print("g4_step done: loss", float(loss.detach()), "rho", crit.downgrade_ratio)


def gen_decoder(UNet):
    """G5: the reference's full UNet (decoder: _UpConv / skip-concat blocks / _Deconv_1x1,
    semi_seg/arch/unet.py:85-97,193-230) in train mode on the G3 'small' weights: decoder outputs at every `until`,
    and one fine-tune step's gradients / BN buffers under the supervised loss of new_epocher.py:270-271
    (KL_div(softmax(logits), onehot) of the un-vendored deepclustering2, restated as -mean log(p_target + 1e-16))."""
    from oracle.spcl_oracle import init_unet_state
    out = {}
    sd = init_unet_state(1, 4, 128, seed=11)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(4, 1, 32, 32, generator=g)  # == G3 small/x
    labels = torch.randint(0, 4, (4, 32, 32), generator=torch.Generator().manual_seed(13))
    out["x"], out["labels"] = x.numpy(), labels.numpy()
    out["param_checksum"] = np.float64(sum(float(v.double().sum()) for v in sd.values()))
    for until in ("Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"):
        ref = UNet(input_dim=1, num_classes=4, max_channel=128)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        out[f"out/{until}"] = ref(x, until=until).detach().numpy()
    net = UNet(input_dim=1, num_classes=4, max_channel=128)
    net.load_state_dict(sd, strict=True)
    net.train()
    logits = net(x)
    out["out/logits"] = logits.detach().numpy()
    prob = logits.softmax(1)
    onehot = torch.nn.functional.one_hot(labels, 4).permute(0, 3, 1, 2).float()
    loss = -(onehot * torch.log(prob + 1e-16)).sum(1).mean()
    out["loss"] = np.float64(loss.item())
    loss.backward()
    for k, p in net.named_parameters():
        out[f"grad/{k}"] = p.grad.numpy()
    for k, b in net.named_buffers():
        out[f"buf/{k}"] = b.numpy()
    net.eval()
    with torch.no_grad():
        ev = net(x)
    out["eval/logits"] = ev.numpy()
    out["eval/pred"] = ev.max(1)[1].numpy()
    np.savez_compressed(os.path.join(OUT, "g5_decoder.npz"), **out)
    print("g5_decoder done")


def gen_round2(SupConLoss1):
    """round 2 additions: SupConLoss1(exclude_other_pos=True), ProjectionHead(pool_name="adaptive_max"),
    DenseProjectionHead (contrastyou/projectors/heads.py:96-120)"""
    from contrastyou.projectors.heads import ProjectionHead, DenseProjectionHead
    out = {}
    cases = []
    for (n, d) in [(4, 16), (8, 256), (30, 256)]:
        z1, z2 = unit_rows(n, d, 100 + n), unit_rows(n, d, 200 + n)
        for lname, labels in label_sets(n).items():
            key = f"xpos/n{n}_d{d}_{lname}"
            a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
            crit = SupConLoss1(temperature=0.07, exclude_other_pos=True)
            loss = crit(a, b, target=labels)
            loss.backward()
            out[f"{key}/loss"], out[f"{key}/dz1"], out[f"{key}/dz2"] = loss.detach().numpy(), a.grad.numpy(), b.grad.numpy()
            cases.append(key)
        out[f"xpos/n{n}_d{d}/z1"], out[f"xpos/n{n}_d{d}/z2"] = z1.numpy(), z2.numpy()
    out["xpos/cases"] = np.array(cases)
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(606)  # the heads below draw their initial weights from the global generator: reproducible fixture
    # adaptive-max pooled head
    head = ProjectionHead(input_dim=32, hidden_dim=24, output_dim=16, head_type="mlp", normalize=True,
                          pool_name="adaptive_max")
    x = torch.randn(5, 32, 7, 9, generator=g, requires_grad=True)
    r = torch.randn(5, 16, generator=g)
    z = head(x)
    (z * r).sum().backward()
    out["maxhead/x"], out["maxhead/r"], out["maxhead/z"], out["maxhead/dx"] = x.detach().numpy(), r.numpy(), z.detach().numpy(), x.grad.numpy()
    for k, p in head.named_parameters():
        out[f"maxhead/param/{k}"], out[f"maxhead/grad/{k}"] = p.detach().numpy(), p.grad.numpy()
    # dense heads: mlp with avg pooling to (5, 4) from 14 x 11 (overlapping, uneven windows), linear with max pooling
    for tag, kw in {"dense_mlp": dict(head_type="mlp", pool_name="adaptive_avg", spatial_size=(5, 4), hidden_dim=24),
                    "dense_lin": dict(head_type="linear", pool_name="adaptive_max", spatial_size=(3, 3), hidden_dim=24)}.items():
        head = DenseProjectionHead(input_dim=16, output_dim=12, normalize=True, **kw)
        x = torch.randn(3, 16, 14, 11, generator=g, requires_grad=True)
        z = head(x)
        r = torch.randn(z.shape, generator=g)
        (z * r).sum().backward()
        out[f"{tag}/x"], out[f"{tag}/r"], out[f"{tag}/z"], out[f"{tag}/dx"] = x.detach().numpy(), r.numpy(), z.detach().numpy(), x.grad.numpy()
        for k, p in head.named_parameters():
            out[f"{tag}/param/{k}"], out[f"{tag}/grad/{k}"] = p.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g6_round2.npz"), **out)
    print("g6_round2:", len(cases), "xpos cases + 3 heads")


def gen_wide(SupConLoss1, SelfPacedSupConLoss):
    """round 3: projection widths beyond 256 (the reference's ``ProjectionHead(output_dim=...)`` and
    ``contrast_loss3.py:25-31`` take any width): both losses at d = 512 and d = 600 (not a multiple of the 256-feature
    chunks), and a ProjectionHead with a 512-wide output."""
    from contrastyou.projectors.heads import ProjectionHead
    out, cases = {}, []
    for (n, d) in [(8, 512), (30, 600)]:
        z1, z2 = unit_rows(n, d, 300 + n), unit_rows(n, d, 400 + n)
        out[f"wide/n{n}_d{d}/z1"], out[f"wide/n{n}_d{d}/z2"] = z1.numpy(), z2.numpy()
        for lname, labels in label_sets(n).items():
            for mname, mode, gamma, cg in LOSS_MODES:
                # (the full grid lives in g1; here a sample that touches every mode and label set -- the fixture stays small)
                if n == 8 and lname in ("distinct", "acdc") and mname not in ("supcon1", "soft_12_cg"):
                    continue
                if n == 30 and (lname, mname) not in (("mod3", "soft_12_cg"), ("none", "supcon1"), ("acdc", "hard_7"),
                                                      ("mod3", "soft_3_cg")):
                    continue
                key = f"wide/n{n}_d{d}_{lname}_{mname}"
                a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                if mode is None:
                    crit = SupConLoss1(temperature=0.07)
                else:
                    crit = SelfPacedSupConLoss(temperature=0.07, weight_update=mode, correct_grad=cg)
                    crit.set_gamma(gamma)
                loss = crit(a, b, target=labels)
                loss.backward()
                out[f"{key}/loss"], out[f"{key}/dz1"], out[f"{key}/dz2"] = loss.detach().numpy(), a.grad.numpy(), b.grad.numpy()
                out[f"{key}/rho"] = np.float32(getattr(crit, "downgrade_ratio", 1.0))
                cases.append(key)
    out["wide/cases"] = np.array(cases)
    torch.manual_seed(707)
    g = torch.Generator().manual_seed(78)
    head = ProjectionHead(input_dim=32, hidden_dim=24, output_dim=512, head_type="mlp", normalize=True)
    x = torch.randn(6, 32, 5, 7, generator=g, requires_grad=True)
    r = torch.randn(6, 512, generator=g)
    z = head(x)
    (z * r).sum().backward()
    out["widehead/x"], out["widehead/r"], out["widehead/z"], out["widehead/dx"] = x.detach().numpy(), r.numpy(), z.detach().numpy(), x.grad.numpy()
    for k, p in head.named_parameters():
        out[f"widehead/param/{k}"], out[f"widehead/grad/{k}"] = p.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g8_wide.npz"), **out)
    print("g8_wide:", len(cases), "loss cases + 1 head")


def gen_data():
    """round 3, SURVEY row N2: index streams of the reference's ``ContrastBatchSampler`` (semi_seg/data/rearr.py:37-98,
    imported BY FILE PATH: it needs only the standard library and torch.utils.data.Sampler) and the partition tables of
    the reference's ``ACDCDataset._get_partition`` / ``ProstateDataset._get_partition`` (semi_seg/data/dataset.py:34-43,
    66-71; the module is executed with its ``contrastyou.*`` imports stubbed -- the base classes are PNG-folder readers --
    and the two methods are called on bare instances that carry only the scan-length table and the ``group_re`` search
    of contrastyou/data/dataset/{acdc.py:16,prostate.py} ).  Inputs: the file stems of the mirror's synthetic stores."""
    import importlib.util
    import random
    import re
    import warnings
    spec = importlib.util.spec_from_file_location("ref_rearr", os.path.join(REF_SRC, "semi_seg", "data", "rearr.py"))
    rearr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rearr)

    # ---- semi_seg/data/dataset.py with stubbed imports
    class _Base:
        group_re = None

        def _get_scan_name(self, filename=None, stem=None):  # contrastyou/data/dataset/base.py:179-186
            return re.compile(self.group_re).search(stem if stem is not None else filename).group(0)

    saved = {k: sys.modules.get(k) for k in ("contrastyou.augment", "contrastyou.data", "contrastyou.data.dataset",
                                             "contrastyou.data.dataset.base", "refdata", "refdata.rearr")}
    def _module(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        return m

    def mk(name, **a):
        sys.modules[name] = _module(name, **a)

    names = ("ACDCDataset", "ProstateDataset", "mmWHSCTDataset", "mmWHSMRDataset", "ProstateMDDataset")
    if "contrastyou" not in sys.modules:
        sys.modules["contrastyou"] = _module("contrastyou")
    mk("contrastyou.augment", SequentialWrapper=object)
    mk("contrastyou.data", **{n: type(n, (_Base,), {}) for n in names})
    mk("contrastyou.data.dataset")
    mk("contrastyou.data.dataset.base", get_stem=lambda p: os.path.splitext(os.path.basename(str(p)))[0])
    pkg = _module("refdata")
    pkg.__path__ = []
    sys.modules["refdata"], sys.modules["refdata.rearr"] = pkg, rearr
    spec = importlib.util.spec_from_file_location("refdata.dataset", os.path.join(REF_SRC, "semi_seg", "data", "dataset.py"))
    refds = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(refds)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v

    import spcl_amd  # noqa: F401  (the mirror only supplies the synthetic file stems)
    from spcl_amd.semi_seg.data import synthetic_slice_store
    out = {}
    for kind, cls, regex, info_attr in (("acdc", refds.ACDCDataset, r"patient\d+_\d+", "_acdc_info"),
                                        ("prostate", refds.ProstateDataset, r"Case\d+", "_prostate_info")):
        store = synthetic_slice_store(scans=7, slices_per_scan=(4, 11) if kind == "acdc" else (9, 26), size=8,
                                      device="cpu", seed=3, kind=kind)
        stems = list(store.get_memory_dictionary()["img"])
        ds = cls.__new__(cls)  # a bare instance: no PNG folder
        ds.group_re = regex
        info = {}
        for f in stems:
            sname = ds._get_scan_name(f)
            info[sname] = max(info.get(sname, 0), int(re.findall(r"\d+", f)[-1]) + 1)
        setattr(ds, info_attr, info)
        ds.get_memory_dictionary = lambda stems=stems: {"img": list(stems)}
        out[f"{kind}/stems"] = np.array(stems)
        out[f"{kind}/partitions"] = np.array([ds._get_partition(f) for f in stems])
        out[f"{kind}/scans"] = np.array([ds._get_scan_name(f) for f in stems])
        out[f"{kind}/scan_len_names"] = np.array(list(info.keys()))
        out[f"{kind}/scan_len"] = np.array(list(info.values()), dtype=np.int64)
        settings = [(3, 1, False), (5, 1, True), (2, 2, False), (6, 3, True)]
        out[f"{kind}/settings"] = np.array([[a, b, int(c)] for a, b, c in settings], dtype=np.int64)
        for si, (scan_num, part_num, shuffle) in enumerate(settings):
            random.seed(11)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", DeprecationWarning)  # random.sample over dict keys (python < 3.11)
                it = iter(rearr.ContrastBatchSampler(ds, scan_sample_num=scan_num, partition_sample_num=part_num,
                                                     shuffle=shuffle))
                batches = [next(it) for _ in range(25)]
            out[f"{kind}/stream{si}/flat"] = np.array([i for b in batches for i in b], dtype=np.int64)
            out[f"{kind}/stream{si}/lens"] = np.array([len(b) for b in batches], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g7_data.npz"), **out)
    print("g7_data:", {k: v.shape for k, v in out.items() if k.endswith("lens") or k.endswith("stems")})


def gen_augment():
    """g9_augment.npz: views PIL ITSELF produces with the operations torchvision's transforms of
    ``ACDCStrongTransforms.pretrain`` (semi_seg/augment.py:6-22) forward to on an 8-bit 'L' image -- torchvision is not
    installed here, and each of these transforms is one PIL call (torchvision/transforms/_functional_pil.py):
    RandomRotation -> ``Image.rotate(angle, NEAREST, expand=False, fillcolor=0)``, RandomVerticalFlip / HorizontalFlip ->
    ``transpose(FLIP_TOP_BOTTOM / FLIP_LEFT_RIGHT)``, RandomCrop(224) -> ``crop((left, top, left + 224, top + 224))``,
    ColorJitter -> ``ImageEnhance.Brightness(img).enhance(b)`` / ``ImageEnhance.Contrast(img).enhance(c)`` in the drawn
    order (saturation / hue: identity on one channel), ToTensor -> / 255.  Inputs: six seeded uint8 slices (square,
    non-square, crop == image); 60 views with seeded parameters over the recipe's ranges plus hand-picked edge cases
    (angle 0 / 45 / -45, factors 1.0, 0.5, 1.5, a factor < 1: PIL's interpolating blend branch, > 1: the clipping one)."""
    import random
    from PIL import Image, ImageEnhance
    rs = np.random.RandomState(90)
    sizes = [(256, 256), (256, 256), (240, 272), (272, 240), (224, 224), (224, 224)]
    slices = []
    for k, (h, w) in enumerate(sizes):
        # smooth structure + noise, full 8-bit range: blobs make a wrong rotation visible, noise makes every pixel count
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        base = 0.5 + 0.25 * np.sin(xx / (7.0 + k)) * np.cos(yy / (11.0 - k)) + 0.25 * rs.rand(h, w)
        slices.append(np.clip(base * 255.0, 0, 255).astype(np.uint8))
    rng = random.Random(91)
    rows = []  # [slice, angle, vflip, hflip, top, left, brightness, contrast, contrast_first]
    for v in range(48):
        si = v % 6
        h, w = sizes[si]
        rows.append([si, rng.uniform(-45.0, 45.0), float(rng.random() < 0.5), float(rng.random() < 0.5),
                     float(rng.randint(0, h - 224)), float(rng.randint(0, w - 224)), rng.uniform(0.5, 1.5),
                     rng.uniform(0.5, 1.5), float(rng.random() < 0.5)])
    edge = [(0.0, 1.0, 1.0), (45.0, 0.5, 1.5), (-45.0, 1.5, 0.5), (10.0, 1.0, 0.75), (-30.0, 0.75, 1.0), (3.25, 1.25, 1.25)]
    for e, (ang, b, c) in enumerate(edge):
        for cf in (0.0, 1.0):
            si = (2 * e + int(cf)) % 6
            h, w = sizes[si]
            rows.append([si, ang, float(e % 2), float((e // 2) % 2), float((h - 224) // 2), float((w - 224) // 3), b, c, cf])
    outs = []
    for si, ang, vf, hf, top, left, b, c, cf in rows:
        im = Image.fromarray(slices[int(si)], "L").rotate(ang, Image.NEAREST, expand=False, fillcolor=0)
        if vf:
            im = im.transpose(Image.FLIP_TOP_BOTTOM)
        if hf:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        top, left = int(top), int(left)
        im = im.crop((left, top, left + 224, top + 224))
        if cf:
            im = ImageEnhance.Brightness(ImageEnhance.Contrast(im).enhance(c)).enhance(b)
        else:
            im = ImageEnhance.Contrast(ImageEnhance.Brightness(im).enhance(b)).enhance(c)
        outs.append(np.asarray(im).copy())
    out = {f"slice{k}": s for k, s in enumerate(slices)}
    out["rows"] = np.array(rows, dtype=np.float64)
    out["views"] = np.stack(outs).astype(np.uint8)
    import PIL
    out["pil_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(OUT, "g9_augment.npz"), **out)
    print("g9_augment:", out["rows"].shape, out["views"].shape, "PIL", PIL.__version__)


def gen_recipes():
    """g10_augment_recipes.npz: views PIL ITSELF produces for the reference's other recipes, and with the interpolation its
    wrapper selects (contrastyou/augment/synchronize.py:95-103: the common transform runs with BILINEAR on images and NEAREST
    on targets; torchvision's RandomRotation / Resize forward to ``Image.rotate(angle, interp)`` / ``Image.resize(size, interp)``):
      acdc_pretrain      rotate(45, BILINEAR) -> flips -> crop(224) -> jitter [0.5, 1.5]      semi_seg/augment.py:6-22
      prostate_pretrain  Resize(224) -> rotate(10, BILINEAR) -> flips -> crop(224, padding=20) -> jitter [0.9, 1.1]   :54-69
      acdc_label         crop(224) -> rotate(30): image BILINEAR, label map NEAREST, no jitter   :23-34
      val                CenterCrop(224)                                                        :35-37
    Inputs: seeded uint8 slices (square and not) with blob label maps of four classes."""
    import random
    from PIL import Image, ImageEnhance, ImageOps
    rs = np.random.RandomState(100)
    sizes = [(256, 256), (240, 272), (288, 256), (320, 320)]
    slices, labels = [], []
    for k, (h, w) in enumerate(sizes):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        base = 0.5 + 0.25 * np.sin(xx / (6.0 + k)) * np.cos(yy / (10.0 - k)) + 0.25 * rs.rand(h, w)
        slices.append(np.clip(base * 255.0, 0, 255).astype(np.uint8))
        coarse = rs.randint(0, 4, size=(h // 16 + 1, w // 16 + 1)).astype(np.uint8)
        labels.append(np.kron(coarse, np.ones((16, 16), dtype=np.uint8))[:h, :w].copy())
    rng = random.Random(101)
    out = {}
    for k in range(len(sizes)):
        out[f"slice{k}"], out[f"label{k}"] = slices[k], labels[k]

    def jitter(im, b, c, cf):
        if cf:
            return ImageEnhance.Brightness(ImageEnhance.Contrast(im).enhance(c)).enhance(b)
        return ImageEnhance.Contrast(ImageEnhance.Brightness(im).enhance(b)).enhance(c)

    # ---- Resize(224) of every slice (torchvision: shorter edge -> 224, BILINEAR)
    def resize224(a):
        h, w = a.shape
        if w <= h:
            ow, oh = 224, int(224 * h / w)
        else:
            oh, ow = 224, int(224 * w / h)
        return np.asarray(Image.fromarray(a, "L").resize((ow, oh), Image.BILINEAR)).copy()
    resized = [resize224(a) for a in slices]
    for k, a in enumerate(resized):
        out[f"resized{k}"] = a
    # ---- acdc_pretrain (bilinear rotation) and prostate_pretrain
    rows_a, views_a, rows_p, views_p = [], [], [], []
    for v in range(24):
        si = v % 4
        h, w = sizes[si]
        ang = [0.0, 45.0, -45.0][v] if v < 3 else rng.uniform(-45.0, 45.0)
        vf, hf = rng.random() < 0.5, rng.random() < 0.5
        top, left = rng.randint(0, h - 224), rng.randint(0, w - 224)
        b, c, cf = rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5), rng.random() < 0.5
        im = Image.fromarray(slices[si], "L").rotate(ang, Image.BILINEAR, expand=False, fillcolor=0)
        if vf:
            im = im.transpose(Image.FLIP_TOP_BOTTOM)
        if hf:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = jitter(im.crop((left, top, left + 224, top + 224)), b, c, cf)
        rows_a.append([si, ang, float(vf), float(hf), top, left, b, c, float(cf)])
        views_a.append(np.asarray(im).copy())
    for v in range(24):
        si = v % 4
        h, w = resized[si].shape
        ang = [0.0, 10.0, -10.0][v] if v < 3 else rng.uniform(-10.0, 10.0)
        vf, hf = rng.random() < 0.5, rng.random() < 0.5
        top, left = rng.randint(0, h + 40 - 224), rng.randint(0, w + 40 - 224)
        b, c, cf = rng.uniform(0.9, 1.1), rng.uniform(0.9, 1.1), rng.random() < 0.5
        im = Image.fromarray(resized[si], "L").rotate(ang, Image.BILINEAR, expand=False, fillcolor=0)
        if vf:
            im = im.transpose(Image.FLIP_TOP_BOTTOM)
        if hf:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = ImageOps.expand(im, border=20, fill=0)  # torchvision F.pad(img, 20, fill=0, 'constant') on a PIL image
        im = jitter(im.crop((left, top, left + 224, top + 224)), b, c, cf)
        rows_p.append([si, ang, float(vf), float(hf), top, left, b, c, float(cf)])
        views_p.append(np.asarray(im).copy())
    # ---- acdc_label: crop, then rotate the crop; image BILINEAR, label NEAREST (the same drawn parameters)
    rows_l, views_l, labs_l = [], [], []
    for v in range(20):
        si = v % 4
        h, w = sizes[si]
        top, left = rng.randint(0, h - 224), rng.randint(0, w - 224)
        ang = [0.0, 30.0, -30.0][v] if v < 3 else rng.uniform(-30.0, 30.0)
        box = (left, top, left + 224, top + 224)
        im = Image.fromarray(slices[si], "L").crop(box).rotate(ang, Image.BILINEAR, expand=False, fillcolor=0)
        lb = Image.fromarray(labels[si], "L").crop(box).rotate(ang, Image.NEAREST, expand=False, fillcolor=0)
        rows_l.append([si, ang, top, left])
        views_l.append(np.asarray(im).copy())
        labs_l.append(np.asarray(lb).copy())
    # ---- val: CenterCrop(224) (torchvision: int(round((h - 224) / 2.0)))
    vals, vlabs = [], []
    for si, (h, w) in enumerate(sizes):
        top, left = int(round((h - 224) / 2.0)), int(round((w - 224) / 2.0))
        vals.append(slices[si][top:top + 224, left:left + 224].copy())
        vlabs.append(labels[si][top:top + 224, left:left + 224].copy())
    out.update(rows_acdc=np.array(rows_a, dtype=np.float64), views_acdc=np.stack(views_a),
               rows_prostate=np.array(rows_p, dtype=np.float64), views_prostate=np.stack(views_p),
               rows_label=np.array(rows_l, dtype=np.float64), views_label=np.stack(views_l), labels_label=np.stack(labs_l),
               views_val=np.stack(vals), labels_val=np.stack(vlabs))
    import PIL
    out["pil_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(OUT, "g10_augment_recipes.npz"), **out)
    print("g10_augment_recipes:", {k: v.shape for k, v in out.items() if k.startswith(("views", "labels_"))}, "PIL", PIL.__version__)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["recipes"]:  # only the round-5 recipe fixture (PIL alone, no reference import)
        gen_recipes()
        return
    if sys.argv[1:] == ["augment"]:  # only the round-4 augmentation fixture (PIL alone, no reference import)
        return gen_augment()
    SupConLoss1, SelfPacedSupConLoss, ProjectionHead, UNet, SFE = _import_reference()
    if sys.argv[1:] == ["round2"]:  # only the round-2 additions (the others are unchanged)
        return gen_round2(SupConLoss1)
    if sys.argv[1:] == ["data"]:  # only the round-3 data-path fixture
        return gen_data()
    if sys.argv[1:] == ["wide"]:  # only the round-3 wide-projection fixture
        return gen_wide(SupConLoss1, SelfPacedSupConLoss)
    gen_loss(SupConLoss1, SelfPacedSupConLoss)
    gen_projector(ProjectionHead)
    gen_encoder(UNet)
    gen_step(UNet, ProjectionHead, SelfPacedSupConLoss, SFE)
    gen_decoder(UNet)
    gen_round2(SupConLoss1)
    gen_data()
    gen_wide(SupConLoss1, SelfPacedSupConLoss)
    gen_augment()


if __name__ == "__main__":
    main()
