"""SURVEY row N2 on the GPU: the augmentation launch against the oracle (geometry bit-exact, colour 2e-6), the device
loader's batch format, and the label generators on real batch meta-data."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spcl_oracle as O


def _store(**kw):
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.data import synthetic_slice_store
    return synthetic_slice_store(device="cuda", **kw)


@pytest.mark.parametrize("size,out", [(256, 224), (64, 64), (97, 50)])
def test_augment_views_geometry_is_bit_exact(size, out):
    from spcl_amd.semi_seg.data import PretrainViews, draw_view_params
    store = _store(scans=3, slices_per_scan=(4, 5), size=size, seed=2)
    views = PretrainViews(store.images, (out, out))
    rng = random.Random(size)
    rows = [draw_view_params(rng.randrange(len(store)), (size, size), (out, out), brightness=None, contrast=None, rng=rng)
            for _ in range(24)]
    rows.append([0, 65536, 0, 0, 0, 0, rows[0][6], rows[0][7]])  # identity: the top-left crop of slice 0
    got = views.apply(rows).cpu().numpy()
    imgs = store.images.cpu().numpy()
    for k, r in enumerate(rows):
        np.testing.assert_array_equal(got[k, 0], O.augment_view(imgs[r[0]], r, (out, out)), err_msg=str(r))
    np.testing.assert_array_equal(got[-1, 0], imgs[0][:out, :out])
    # a quarter turn of the full image is a transpose + flip, exactly
    if out == size:
        q = views.apply([[1, 0, 65536, 0, 0, 0, rows[0][6], rows[0][7]]]).cpu().numpy()[0, 0]
        want = O.augment_view(imgs[1], [1, 0, 65536, 0, 0, 0, rows[0][6], rows[0][7]], (out, out))
        np.testing.assert_array_equal(q, want)
        assert np.array_equal(q, np.rot90(imgs[1], 1)) or np.array_equal(q, np.rot90(imgs[1], -1))


def test_augment_views_colour_and_two_independent_views():
    from spcl_amd.semi_seg.data import PretrainViews
    store = _store(scans=4, slices_per_scan=(6, 8), size=256, seed=4)
    views = PretrainViews(store.images, (224, 224), pil_exact=False)  # the round-2 float recipe (8-int rows)
    rng = random.Random(9)
    idx = [3, 7, 11, 20, 5]
    rows = views.params(idx, rng)
    assert len(rows) == 10 and [r[0] for r in rows] == idx + idx and rows[0] != rows[5]  # total_freedom: own draws per view
    got = views.apply(rows).cpu().numpy()
    imgs = store.images.cpu().numpy()
    for k, r in enumerate(rows):
        want = O.augment_view(imgs[r[0]], r, (224, 224))
        np.testing.assert_allclose(got[k, 0], want, atol=2e-6, rtol=0)
    assert got.min() >= 0.0 and got.max() <= 1.0
    a, b = views(idx, random.Random(1))
    assert a.shape == b.shape == (5, 1, 224, 224) and a.is_cuda and not torch.equal(a, b)


def _g9():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_augment.npz"))


def _g9_row(r, hw):
    """a golden row [slice, angle, vflip, hflip, top, left, b, c, contrast_first] as spcl_augment_views_pil takes it"""
    import struct
    bits = lambda x: struct.unpack("<i", struct.pack("<f", float(x)))[0]  # noqa: E731
    si, ang, vf, hf, top, left, b, c, cf = r
    flags = (1 if hf else 0) | (2 if vf else 0) | (4 if cf else 0)
    return [int(si)] + O.pil_affine_q16(float(ang), hw[1], hw[0]) + [flags, int(top), int(left), bits(b), bits(c)]


def test_augment_views_pil_reproduces_what_pil_itself_wrote():
    """`spcl_augment_views_pil` against tests/golden/g9_augment.npz -- 60 views PIL 12.2 produced with the calls torchvision's
    transforms of ``ACDCStrongTransforms.pretrain`` (semi_seg/augment.py:6-22) forward to: rotation (nearest, PIL's fixed-point
    affine), both flips, crop, brightness / contrast in both orders, ToTensor.  EVERY pixel of every view, bit for bit; and the
    oracle's restatement gives the same bits (the CPU suite pins it to the fixture as well)."""
    from spcl_amd.semi_seg.data import PretrainViews
    g = _g9()
    rows, want = g["rows"], g["views"]
    for si in range(6):
        img = g[f"slice{si}"]
        store = torch.from_numpy(img.astype(np.float32) / np.float32(255)).cuda()[None]  # 8-bit grey levels as k / 255
        views = PretrainViews(store, (224, 224))
        sel = [k for k in range(len(rows)) if int(rows[k][0]) == si]
        prm = [[0] + _g9_row(rows[k], img.shape)[1:] for k in sel]
        got = views.apply(prm).cpu().numpy()[:, 0]
        for j, k in enumerate(sel):
            np.testing.assert_array_equal(got[j], want[k].astype(np.float32) / np.float32(255), err_msg=str(rows[k]))
            np.testing.assert_array_equal(got[j], O.augment_view_pil(img, [0] + prm[j][1:], (224, 224)))


def test_augment_views_pil_random_rows_match_the_oracle():
    """the product's own parameter draws (``PretrainViews.params``: 12-int rows in torchvision's draw order) on a synthetic
    store, odd sizes included: kernel == oracle.augment_view_pil on every pixel; two independent views per slice"""
    from spcl_amd.semi_seg.data import PretrainViews
    for size, out in ((256, 224), (97, 50), (64, 64)):
        store = _store(scans=3, slices_per_scan=(4, 5), size=size, seed=2)
        u8 = torch.round(store.images * 255).clamp(0, 255)
        store_q = (u8 / 255).contiguous()  # an 8-bit store, as decoded PNG slices are
        views = PretrainViews(store_q, (out, out))
        rng = random.Random(size)
        idx = [rng.randrange(store_q.shape[0]) for _ in range(6)]
        rows = views.params(idx, rng)
        assert len(rows) == 12 and all(len(r) == 12 for r in rows) and rows[0] != rows[6]
        got = views.apply(rows).cpu().numpy()[:, 0]
        imgs = u8.cpu().numpy().astype(np.uint8)
        for k, r in enumerate(rows):
            np.testing.assert_array_equal(got[k], O.augment_view_pil(imgs[r[0]], r, (out, out)), err_msg=str(r))


def test_contrastive_device_loader_batches():
    from spcl_amd.semi_seg.data import ProstateSliceStore, get_contrastive_dataloader
    from spcl_amd.semi_seg.hooks.utils import get_label
    store = _store(scans=14, slices_per_scan=(9, 12), size=256, seed=6)
    loader, _ = get_contrastive_dataloader(store, {"scan_sample_num": 10, "partition_sample_num": 1, "num_workers": 8})
    random.seed(21)
    (img, img_tf, tgt, tgt_tf), filenames, (partitions, scans) = next(iter(loader))
    assert img.shape == img_tf.shape == (30, 1, 224, 224) and img.is_cuda  # config/pretrain.yaml:14-17: 10 scans x 3
    assert len(set(scans)) == 10 and sorted(set(partitions)) == ["0", "1", "2"]
    assert all(f.startswith(s) for f, s in zip(filenames, scans))
    labels = get_label(contrast_on="partition", data_name="acdc", partition_group=partitions, label_group=scans)
    assert labels == O.get_label("partition", "acdc", partitions, scans) and set(labels) == {0, 1, 2}
    random.seed(21)
    again = next(iter(get_contrastive_dataloader(store, {"scan_sample_num": 10, "partition_sample_num": 1})[0]))
    assert again[1] == filenames and torch.equal(again[0][0], img)  # same python seed -> same batch, same views
    # other data sets: infinite random permutation, batch = scan_sample_num x partition_num (_helper.py:52-53)
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.data import synthetic_slice_store
    pstore = synthetic_slice_store(scans=5, slices_per_scan=(16, 20), size=224, device="cuda", kind="prostate")
    assert isinstance(pstore, ProstateSliceStore)
    ploader, _ = get_contrastive_dataloader(pstore, {"scan_sample_num": 2, "partition_sample_num": 1})
    (pimg, _, _, _), pf, (pp, ps) = next(iter(ploader))
    assert pimg.shape == (16, 1, 224, 224) and all(s.startswith("Case") for s in ps)
    with pytest.raises(TypeError):
        get_contrastive_dataloader([1, 2, 3], {"scan_sample_num": 2})


def test_flip_pair_equals_cat_of_view1_and_flipped_view2():
    """TensorRandomFlip.apply_pair (spcl_flip_pair: the pre-train step's input pair in one launch) == torch.cat([first,
    apply_batch(second)]) with the same random stream, for f32 and bf16, vector and scalar widths."""
    import random

    import spcl_amd  # noqa
    from spcl_amd.semi_seg.epochers.helper import TensorRandomFlip
    for dtype, W in ((torch.float32, 224), (torch.bfloat16, 64), (torch.float32, 30)):
        g = torch.Generator().manual_seed(W)
        a = torch.rand(6, 1, 20, W, generator=g).cuda().to(dtype)
        b = torch.rand(6, 1, 20, W, generator=g).cuda().to(dtype)
        tf = TensorRandomFlip(axis=[1, 2], threshold=0.5)
        random.seed(5)
        pair = tf.apply_pair(a, b)
        random.seed(5)
        ref = torch.cat([a, tf.apply_batch(b)], dim=0)
        assert pair.shape == ref.shape and torch.equal(pair, ref)
        assert not torch.equal(pair[6:], b)  # some sample really was flipped


# ---- round 5: spcl_augment_views_recipe / spcl_resize_bilinear_pil against what PIL itself wrote (g10)
def _g10():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_augment_recipes.npz"))


def _dev_store(a_u8):
    return torch.from_numpy(a_u8.astype(np.float32) / np.float32(255)).cuda()[None]


def test_recipe_views_reproduce_what_pil_itself_wrote():
    """ACDC pre-train views with the BILINEAR image rotation, Prostate pre-train views (device Resize(224) + rotation + flips +
    RandomCrop(224, padding=20) + jitter), ACDC labelled pairs (crop, then rotation; image BILINEAR, label map NEAREST -> int64)
    and CenterCrop(224): EVERY pixel of every view of tests/golden/g10_augment_recipes.npz, bit for bit."""
    from spcl_amd.semi_seg.data import augment as A
    g = _g10()
    for k in range(4):  # the store-build Resize(224): bilinear in Resample.c's fixed point, two passes
        got = A.resize_store(_dev_store(g[f"slice{k}"]), 224)
        np.testing.assert_array_equal(torch.round(got[0] * 255).cpu().numpy().astype(np.uint8), g[f"resized{k}"])
        np.testing.assert_array_equal(got[0].cpu().numpy(), g[f"resized{k}"].astype(np.float32) / np.float32(255))
    for r, w in zip(g["rows_acdc"], g["views_acdc"]):
        si, ang, vf, hf, top, left, b, c, cf = r
        img = g[f"slice{int(si)}"]
        views = A.RecipeViews(_dev_store(img), "acdc_pretrain", (224, 224))
        row = A.recipe_row(0, img.shape, (224, 224), angle=float(ang), vflip=bool(vf), hflip=bool(hf), top=int(top), left=int(left),
                           brightness=b, contrast=c, contrast_first=bool(cf))
        got = views.apply([row]).cpu().numpy()[0, 0]
        np.testing.assert_array_equal(got, w.astype(np.float32) / np.float32(255), err_msg=str(r))
    for r, w in zip(g["rows_prostate"], g["views_prostate"]):
        si, ang, vf, hf, top, left, b, c, cf = r
        views = A.RecipeViews(_dev_store(g[f"slice{int(si)}"]), "prostate_pretrain", (224, 224))  # (resizes the store itself)
        assert tuple(views.images.shape[1:]) == g[f"resized{int(si)}"].shape
        row = A.recipe_row(0, tuple(views.images.shape[1:]), (224, 224), angle=float(ang), vflip=bool(vf), hflip=bool(hf),
                           top=int(top), left=int(left), pad=20, brightness=b, contrast=c, contrast_first=bool(cf))
        got = views.apply([row]).cpu().numpy()[0, 0]
        np.testing.assert_array_equal(got, w.astype(np.float32) / np.float32(255), err_msg=str(r))
    for r, w, lw in zip(g["rows_label"], g["views_label"], g["labels_label"]):
        si, ang, top, left = r
        img, lab = g[f"slice{int(si)}"], g[f"label{int(si)}"]
        views = A.RecipeViews(_dev_store(img), "acdc_label", (224, 224), labels=torch.from_numpy(lab).cuda()[None])
        row = A.recipe_row(0, img.shape, (224, 224), angle=float(ang), top=int(top), left=int(left), crop_first=True)
        got, glab = views.apply([row], with_labels=True)
        assert glab.dtype == torch.int64 and tuple(glab.shape) == (1, 1, 224, 224)
        np.testing.assert_array_equal(got.cpu().numpy()[0, 0], w.astype(np.float32) / np.float32(255), err_msg=str(r))
        np.testing.assert_array_equal(glab.cpu().numpy()[0, 0], lw.astype(np.int64), err_msg=str(r))
    for si in range(4):
        views = A.RecipeViews(_dev_store(g[f"slice{si}"]), "acdc_label", (224, 224),
                              labels=torch.from_numpy(g[f"label{si}"]).cuda()[None])
        got, glab = views.val([0])
        np.testing.assert_array_equal(torch.round(got[0, 0] * 255).cpu().numpy().astype(np.uint8), g["views_val"][si])
        np.testing.assert_array_equal(glab.cpu().numpy()[0, 0], g["labels_val"][si].astype(np.int64))


def test_recipe_views_random_rows_match_the_oracle_and_pairs_are_independent():
    """the product's own draws for every recipe on a synthetic 8-bit store (odd sizes too): kernel == oracle.recipe_view on
    every pixel; ``pairs`` gives two different views per slice, ``labelled`` one geometry for image and label map"""
    import struct
    from spcl_amd.semi_seg.data import augment as A
    f32 = lambda i: struct.unpack("<f", struct.pack("<i", i))[0]  # noqa: E731
    f64 = lambda lo, hi: struct.unpack("<d", struct.pack("<ii", lo, hi))[0]  # noqa: E731
    # (the last one: RandomCrop(size, padding=20) THEN RandomRotation -- Spleen `label`, semi_seg/augment.py:107-112 -- the
    # crop window reaches into the zero padding around the slice, on the last slice of the store too: ADVICE r05)
    padded_label = dict(degrees=10.0, flips=False, pad=20, crop_first=True, brightness=None, contrast=None, resize=None)
    for name, size, out in (("acdc_pretrain", 256, 224), ("prostate_pretrain", 250, 224), ("acdc_label", 97, 50),
                            (padded_label, 64, 64)):
        store = _store(scans=3, slices_per_scan=(4, 5), size=size, seed=5)
        u8 = torch.round(store.images * 255).clamp(0, 255)
        lab = (u8 / 64).floor().clamp(0, 3).to(torch.uint8) if (isinstance(name, dict) or "label" in name) else None
        views = A.RecipeViews((u8 / 255).contiguous(), name, (out, out), labels=lab)
        imgs = torch.round(views.images * 255).cpu().numpy().astype(np.uint8)
        rng = random.Random(size)
        idx = [rng.randrange(imgs.shape[0]) for _ in range(5)]
        if isinstance(name, dict):
            idx[-1] = imgs.shape[0] - 1  # the store's last slice: nothing behind it to read
        if lab is None:
            a, b = views.pairs(idx, random.Random(3))
            assert tuple(a.shape) == (5, 1, out, out) and not torch.equal(a, b)
        rows = views.rows(idx, rng)
        res = views.apply(rows, with_labels=lab is not None)
        got = (res[0] if lab is not None else res).cpu().numpy()[:, 0]
        for k, r in enumerate(rows):
            fl = r[1]
            # (the angle is not in the row: the oracle gets PIL's matrix back through atan2 of its doubles -- exact enough to
            # reproduce round(cos, 15) / round(sin, 15) -- so compare through the oracle's own matrix entry points instead)
            ang = -np.degrees(np.arctan2(f64(r[16], r[17]), f64(r[14], r[15])))
            want, wlab = O.recipe_view(imgs[r[0]], lab[r[0]].cpu().numpy() if lab is not None else None, (out, out), angle=float(ang),
                                       vflip=bool(fl & 2), hflip=bool(fl & 1), top=r[2], left=r[3], pad=r[4],
                                       crop_first=bool(fl & 16), brightness=f32(r[5]), contrast=f32(r[6]), contrast_first=bool(fl & 4))
            if O.pil_affine_q16(float(ang), *(((out, out)) if fl & 16 else (imgs.shape[2], imgs.shape[1]))) != r[8:14]:
                continue  # (the recovered angle differs from the drawn one in its last bit: another matrix)
            np.testing.assert_array_equal(got[k], want.astype(np.float32) / np.float32(255), err_msg=f"{name} {r[:8]}")
            if lab is not None:
                np.testing.assert_array_equal(res[1].cpu().numpy()[k, 0], wlab.astype(np.int64))


def test_labelled_loader_and_default_pretrain_recipe():
    """LabeledDeviceLoader: batches in the reference's tuple format from a store with label maps, image and label map through
    ONE geometry (label pixels are the store's class codes or the rotation's zero fill); the contrastive loader's default views
    are the data set's own pre-train recipe (bilinear image rotation; Prostate: resized to 224 when the store is built)"""
    from spcl_amd.semi_seg.data import ContrastiveDeviceLoader, InfiniteRandomSampler, LabeledDeviceLoader, RecipeViews
    store = _store(scans=4, slices_per_scan=(6, 8), size=256, seed=8)
    u8 = torch.round(store.images * 255).clamp(0, 255)
    store.images = (u8 / 255).contiguous()
    store.targets = (u8 / 64).floor().clamp(0, 3).to(torch.uint8)
    random.seed(5)
    (img, img2, tgt, tgt2), names, (parts, scans) = next(iter(LabeledDeviceLoader(store, batch_size=6)))
    assert tuple(img.shape) == (6, 1, 224, 224) and tgt.dtype == torch.int64 and tuple(tgt.shape) == (6, 1, 224, 224)
    # SequentialWrapperTwice keeps total_freedom=True for the `label` recipes (semi_seg/augment.py:23-34): the second pair is an
    # independent crop / rotation of the same slices
    assert not torch.equal(img2, img) and not torch.equal(tgt2, tgt) and tuple(img2.shape) == tuple(img.shape)
    assert len(names) == 6 and all(n.startswith(s) for n, s in zip(names, scans))
    assert int(tgt.min()) >= 0 and int(tgt.max()) <= 3
    # total_freedom=False: one geometry; a recipe without image-only randomness then returns the same tensors twice, the
    # `pretrain` recipe (what creator.get_data trains on, creator.py:30) a fresh colour jitter of the same geometry
    (i1, i2, t1, t2), _, _ = next(iter(LabeledDeviceLoader(store, batch_size=4, total_freedom=False)))
    assert torch.equal(i1, i2) and torch.equal(t1, t2)
    (i1, i2, t1, t2), _, _ = next(iter(LabeledDeviceLoader(store, batch_size=4, recipe="acdc_pretrain", total_freedom=False)))
    assert torch.equal(t1, t2) and not torch.equal(i1, i2) and tuple(i1.shape) == (4, 1, 224, 224)
    # label = floor(level / 64) in the store; both went through one geometry, the image bilinearly: away from class borders
    # (where the four neighbours agree) the relation still holds
    lv = torch.round(img * 255)
    agree = ((lv / 64).floor().clamp(0, 3).long() == tgt).float().mean()
    assert float(agree) > 0.9
    with pytest.raises(ValueError):
        LabeledDeviceLoader(_store(scans=2, slices_per_scan=(4, 5), size=64, seed=1), batch_size=2)
    loader = ContrastiveDeviceLoader(store, sampler=InfiniteRandomSampler(store), batch_size=4)
    assert isinstance(loader._views, RecipeViews) and loader._views.recipe["degrees"] == 45.0
    (a, b, _, _), _, _ = next(iter(loader))
    assert tuple(a.shape) == (4, 1, 224, 224) and not torch.equal(a, b)


def test_split_recipe_launches_equal_the_one_workgroup_launch():
    """``spcl_augment_views_recipe_ws`` (16 workgroups per view + a finishing launch: what RecipeViews calls since round 6) against
    ``spcl_augment_views_recipe`` (one workgroup per view) on drawn pre-train and labelled rows: every pixel of every view and
    label map, bit for bit -- the partial sums are integers, the contrast step's mean is the same number."""
    from spcl_amd import native as _n
    from spcl_amd.semi_seg.data import augment as A
    store = _store(scans=6, slices_per_scan=(5, 7), size=256, seed=12)
    S, HS, WS = store.images.shape
    labels = (torch.rand(S, HS, WS, device="cuda") * 4).to(torch.uint8)
    rng = random.Random(7)
    for recipe, with_labels, out_hw in (("acdc_pretrain", False, (224, 224)), ("prostate_pretrain", False, (224, 224)),
                                        ("acdc_label", True, (224, 224)), ("acdc_pretrain", False, (97, 131))):
        views = A.RecipeViews(store.images, recipe, out_hw, labels=labels if with_labels else None)
        rows = views.rows([rng.randrange(S) for _ in range(23)], rng)
        got = views.apply(rows, with_labels=with_labels)
        p = A._upload_i32(rows, "cuda")
        oh, ow = out_hw
        ref = torch.empty(len(rows), 1, oh, ow, dtype=torch.float32, device="cuda")
        ref_lab = torch.empty(len(rows), 1, oh, ow, dtype=torch.int64, device="cuda") if with_labels else None
        si, hs, ws = views.images.shape
        _n.call("spcl_augment_views_recipe", _n.ptr(views.images), _n.ptr(views.labels) if with_labels else None, si, hs, ws,
                _n.ptr(p), len(rows), _n.ptr(ref), _n.ptr(ref_lab), oh, ow, int(views.recipe["pad"]), _n.stream())
        if with_labels:
            assert torch.equal(got[0], ref) and torch.equal(got[1], ref_lab), recipe
        else:
            assert torch.equal(got, ref), recipe
        assert float(ref.max()) > 0.0
