"""SURVEY row N2 (on-device data path), host side: the contrastive batch composition, the partition meta-labels and the
augmentation parameter draws against the oracle's literal restatement of the reference (semi_seg/data/rearr.py:37-98,
semi_seg/data/dataset.py:34-43,66-71, semi_seg/augment.py:6-22)."""
import random

import numpy as np

import pytest
import torch

from oracle import spcl_oracle as O

import spcl_amd  # noqa: F401
from spcl_amd.semi_seg.data import (ContrastBatchSampler, InfiniteRandomSampler, acdc_partition, draw_view_params,
                                    prostate_partition, synthetic_slice_store)


@pytest.mark.parametrize("scan_num,part_num,shuffle", [(3, 1, False), (5, 1, True), (2, 2, False), (6, 3, True)])
def test_contrast_batch_sampler_draws_the_reference_batches(scan_num, part_num, shuffle):
    store = synthetic_slice_store(scans=7, slices_per_scan=(4, 11), size=16, device="cpu", seed=3)
    scans, parts = store.show_scan_names(), store.show_partitions()
    random.seed(11)
    it = iter(ContrastBatchSampler(store, scan_sample_num=scan_num, partition_sample_num=part_num, shuffle=shuffle))
    got = [next(it) for _ in range(25)]
    random.seed(11)
    want = [O.contrast_batch_indices(scans, parts, scan_num, part_num, shuffle) for _ in range(25)]
    assert got == want
    for b in got:  # every drawn (scan, partition) pair contributes exactly partition_sample_num slices
        pairs = {}
        for i in b:
            pairs.setdefault((scans[i], parts[i]), []).append(i)
        assert all(len(v) == part_num for v in pairs.values()) and len({s for s, _ in pairs}) <= scan_num


def test_short_scans_skip_partitions_they_cannot_fill():
    store = synthetic_slice_store(scans=4, slices_per_scan=(2, 3), size=8, device="cpu", seed=1)  # 2-3 slices: empty thirds
    it = iter(ContrastBatchSampler(store, scan_sample_num=4, partition_sample_num=1))
    random.seed(5)
    got = next(it)
    random.seed(5)
    assert got == O.contrast_batch_indices(store.show_scan_names(), store.show_partitions(), 4, 1)
    assert len(got) < 4 * 3  # some (scan, partition) pairs are empty
    with pytest.raises(AssertionError):
        iter(ContrastBatchSampler(store, scan_sample_num=5))


def test_partition_meta_labels():
    # ACDC: scan of 10 slices -> cutting point 3: indices 0-2 | 3-6 | 7-9 (dataset.py:34-43)
    assert [acdc_partition(f"patient004_00_{k:02d}", 10) for k in range(10)] == list("0001111222")
    assert [acdc_partition(f"patient100_01_{k}", 7) for k in range(7)] == list("0011122")
    # Prostate: 20 slices, 8 partitions -> cutting point 2 -> index // 3 (dataset.py:66-71)
    assert [prostate_partition(f"Case07_{k:02d}", 20) for k in range(20)] == [str(k // 3) for k in range(20)]
    for k in range(12):
        assert acdc_partition(f"patient001_00_{k:02d}", 12) == O.acdc_partition(f"patient001_00_{k:02d}", 12)
        assert prostate_partition(f"Case01_{k:02d}", 12) == O.prostate_partition(f"Case01_{k:02d}", 12)
    store = synthetic_slice_store(scans=3, slices_per_scan=(9, 9), size=8, device="cpu")
    assert store.meta(4) == ("patient001_00_04", "1", "patient001_00")
    assert store.get_scan_list() == ["patient001_00", "patient002_00", "patient003_00"]
    with pytest.raises(AttributeError):
        store._get_scan_name("scan_without_pattern")


def test_view_parameter_draws_follow_the_recipe():
    rng = random.Random(0)
    rows = [draw_view_params(7, (256, 256), (224, 224), rng=rng) for _ in range(4000)]
    import struct
    f = lambda bits: struct.unpack("<f", struct.pack("<i", bits))[0]  # noqa: E731
    for r in rows:
        assert r[0] == 7 and 0 <= r[4] <= 32 and 0 <= r[5] <= 32 and 0 <= r[3] < 8
        assert abs(r[1] ** 2 + r[2] ** 2 - 65536 ** 2) < 2 * 65536 * 2  # a rotation, quantised to 16.16
        assert r[1] >= int(0.7071 * 65536) - 1  # |angle| <= 45 degrees
        assert 0.5 <= f(r[6]) <= 1.5 and 0.5 <= f(r[7]) <= 1.5
    for bit in (1, 2, 4):  # flips and the jitter order are fair coins
        assert 0.45 < sum(1 for r in rows if r[3] & bit) / len(rows) < 0.55
    same = draw_view_params(0, (224, 224), (224, 224), degrees=0, brightness=None, contrast=None, flips=False, rng=rng)
    assert same[1:3] == [65536, 0] and same[4:6] == [0, 0] and f(same[6]) == 1.0 and f(same[7]) == 1.0


def test_infinite_random_sampler_is_a_stream_of_permutations():
    random.seed(2)
    it = iter(InfiniteRandomSampler(range(5)))
    a, b = [next(it) for _ in range(5)], [next(it) for _ in range(5)]
    assert sorted(a) == sorted(b) == list(range(5))


@pytest.mark.parametrize("kind", ["acdc", "prostate"])
def test_sampler_and_partitions_against_the_reference_fixture(golden, kind):
    """g7_data.npz (tools/gen_golden.py gen_data): index streams drawn by the REFERENCE's ``ContrastBatchSampler``
    (semi_seg/data/rearr.py:37-98, imported by file path) and the partition tables of the reference's
    ``_get_partition`` (semi_seg/data/dataset.py:34-43,66-71) on the synthetic stores' file stems -- the mirror and the
    oracle's restatement both reproduce them exactly."""
    g = golden("g7_data.npz")
    store = synthetic_slice_store(scans=7, slices_per_scan=(4, 11) if kind == "acdc" else (9, 26), size=8, device="cpu",
                                  seed=3, kind=kind)
    stems = [str(s) for s in g[f"{kind}/stems"]]
    assert list(store.get_memory_dictionary()["img"]) == stems
    want_parts, want_scans = [str(p) for p in g[f"{kind}/partitions"]], [str(s) for s in g[f"{kind}/scans"]]
    assert store.show_partitions() == want_parts
    assert store.show_scan_names() == want_scans
    lens = dict(zip((str(s) for s in g[f"{kind}/scan_len_names"]), (int(v) for v in g[f"{kind}/scan_len"])))
    part = O.acdc_partition if kind == "acdc" else O.prostate_partition
    assert [part(f, lens[s]) for f, s in zip(stems, want_scans)] == want_parts  # the oracle's restatement
    assert len(set(want_parts)) == (3 if kind == "acdc" else 8)
    for si, (scan_num, part_num, shuffle) in enumerate(g[f"{kind}/settings"].tolist()):
        flat, ls = g[f"{kind}/stream{si}/flat"].tolist(), g[f"{kind}/stream{si}/lens"].tolist()
        want, off = [], 0
        for n in ls:
            want.append(flat[off:off + n])
            off += n
        random.seed(11)
        it = iter(ContrastBatchSampler(store, scan_sample_num=scan_num, partition_sample_num=part_num,
                                       shuffle=bool(shuffle)))
        assert [next(it) for _ in range(25)] == want, (kind, si)
        random.seed(11)
        assert [O.contrast_batch_indices(want_scans, want_parts, scan_num, part_num, bool(shuffle))
                for _ in range(25)] == want, (kind, si)


def test_oracle_augment_view_pil_is_pinned_to_pil():
    """oracle.augment_view_pil -- the restatement the HIP kernel `spcl_augment_views_pil` is tested against -- reproduces, bit
    for bit, the 60 views of tests/golden/g9_augment.npz that PIL itself produced (tools/gen_golden.py augment: the PIL calls
    torchvision's RandomRotation / flips / RandomCrop / ColorJitter / ToTensor of semi_seg/augment.py:6-22 forward to)."""
    import os
    import struct
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_augment.npz"))
    bits = lambda x: struct.unpack("<i", struct.pack("<f", float(x)))[0]  # noqa: E731
    rows, want = g["rows"], g["views"]
    assert rows.shape == (60, 9) and want.shape == (60, 224, 224) and want.dtype == np.uint8
    seen = set()
    for r, w in zip(rows, want):
        si, ang, vf, hf, top, left, b, c, cf = r
        img = g[f"slice{int(si)}"]
        flags = (1 if hf else 0) | (2 if vf else 0) | (4 if cf else 0)
        row = [0] + O.pil_affine_q16(float(ang), img.shape[1], img.shape[0]) + [flags, int(top), int(left), bits(b), bits(c)]
        got = O.augment_view_pil(img, row, (224, 224))
        np.testing.assert_array_equal(got, w.astype(np.float32) / np.float32(255), err_msg=str(r))
        seen.add((flags, b <= 1.0, c <= 1.0))
    assert len(seen) >= 12  # flips, both jitter orders, interpolating (<= 1) and clipping (> 1) blends all occur


def test_pil_exact_parameter_rows_follow_the_reference_ranges():
    from spcl_amd.semi_seg.data.augment import draw_view_params_pil, pil_affine_q16
    import struct
    import numpy as np
    rng = random.Random(5)
    rows = [draw_view_params_pil(7, (256, 256), (224, 224), rng=rng) for _ in range(2000)]
    f = lambda i: struct.unpack("<f", struct.pack("<i", i))[0]  # noqa: E731
    assert all(r[0] == 7 and len(r) == 12 for r in rows)
    assert all(0 <= r[8] <= 32 and 0 <= r[9] <= 32 for r in rows)                      # RandomCrop(224) of 256
    assert all(0.5 <= f(r[10]) <= 1.5 and 0.5 <= f(r[11]) <= 1.5 for r in rows)        # ColorJitter ranges
    cos45 = int(np.floor(np.cos(np.radians(45.0)) * 65536 + 0.5))
    assert all(cos45 - 1 <= r[1] <= 65536 and abs(r[2]) <= cos45 + 1 and r[4] == -r[2] and r[5] == r[1] for r in rows)  # |angle| <= 45
    for bit in (1, 2, 4):
        share = sum(1 for r in rows if r[7] & bit) / len(rows)
        assert 0.45 < share < 0.55
    assert pil_affine_q16(0.0, 224, 224) == O.pil_affine_q16(0.0, 224, 224) == [65536, 0, 32768, 0, 65536, 32768]
    assert pil_affine_q16(17.5, 272, 240) == O.pil_affine_q16(17.5, 272, 240)



# ---- round 5: the other recipes of semi_seg/augment.py, with the interpolation the reference's wrapper selects
def _g10():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_augment_recipes.npz"))


def test_oracle_recipes_are_pinned_to_pil():
    """oracle.recipe_view / pil_resize_bilinear / pil_rotate_bilinear -- what `spcl_augment_views_recipe` and
    `spcl_resize_bilinear_pil` are tested against -- reproduce, bit for bit, what PIL 12.2 itself wrote into
    tests/golden/g10_augment_recipes.npz (tools/gen_golden.py recipes): ACDC pre-train views with the BILINEAR image rotation
    the reference's wrapper selects (contrastyou/augment/synchronize.py:95-103), Prostate pre-train views (Resize(224), rotation,
    flips, RandomCrop(224, padding=20), jitter), ACDC labelled pairs (crop, then rotation: image BILINEAR, label map NEAREST),
    CenterCrop(224)."""
    g = _g10()
    for k in range(4):
        oh, ow = O.resize_shorter_edge(g[f"slice{k}"].shape, 224)
        assert (oh, ow) == g[f"resized{k}"].shape
        np.testing.assert_array_equal(O.pil_resize_bilinear(g[f"slice{k}"], (oh, ow)), g[f"resized{k}"])
    for r, w in zip(g["rows_acdc"], g["views_acdc"]):
        si, ang, vf, hf, top, left, b, c, cf = r
        got, _ = O.recipe_view(g[f"slice{int(si)}"], None, (224, 224), angle=float(ang), vflip=bool(vf), hflip=bool(hf),
                               top=int(top), left=int(left), brightness=b, contrast=c, contrast_first=bool(cf))
        np.testing.assert_array_equal(got, w, err_msg=str(r))
    for r, w in zip(g["rows_prostate"], g["views_prostate"]):
        si, ang, vf, hf, top, left, b, c, cf = r
        got, _ = O.recipe_view(g[f"resized{int(si)}"], None, (224, 224), angle=float(ang), vflip=bool(vf), hflip=bool(hf),
                               top=int(top), left=int(left), pad=20, brightness=b, contrast=c, contrast_first=bool(cf))
        np.testing.assert_array_equal(got, w, err_msg=str(r))
    for r, w, lw in zip(g["rows_label"], g["views_label"], g["labels_label"]):
        si, ang, top, left = r
        got, lab = O.recipe_view(g[f"slice{int(si)}"], g[f"label{int(si)}"], (224, 224), angle=float(ang), top=int(top),
                                 left=int(left), crop_first=True)
        np.testing.assert_array_equal(got, w, err_msg=str(r))
        np.testing.assert_array_equal(lab, lw, err_msg=str(r))
    # the image of a labelled pair is NOT what a nearest rotation gives (the round-4 recipe rotated images with NEAREST)
    r = g["rows_label"][5]
    crop = g[f"slice{int(r[0])}"][int(r[2]):int(r[2]) + 224, int(r[3]):int(r[3]) + 224]
    assert (O.pil_rotate_nearest(np.ascontiguousarray(crop), float(r[1])) != g["views_label"][5]).mean() > 0.3


def test_oracle_padded_crop_first_is_pinned_to_pil():
    """RandomCrop(size, padding=20) followed by RandomRotation (Spleen `label`, semi_seg/augment.py:107-112): the oracle's
    crop-first path pads with zeros before it crops, as torchvision does (F.pad on the PIL image, then F.crop, then
    Image.rotate: BILINEAR for the image, NEAREST for the label map) -- checked against PIL itself, here"""
    from PIL import Image, ImageOps
    rng = np.random.RandomState(4)
    img = (rng.rand(64, 72) * 255).astype(np.uint8)
    lab = (img // 64).astype(np.uint8)
    for top, left, ang in ((0, 0, 7.5), (40, 48, -9.0), (13, 29, 3.25), (40, 0, 0.0)):
        want, wlab = [], []
        for a, res in ((img, Image.BILINEAR), (lab, Image.NEAREST)):
            im = ImageOps.expand(Image.fromarray(a), border=20, fill=0).crop((left, top, left + 64, top + 64))
            (want if res == Image.BILINEAR else wlab).append(np.asarray(im.rotate(ang, res)))
        got, glab = O.recipe_view(img, lab, (64, 64), angle=ang, top=top, left=left, pad=20, crop_first=True)
        np.testing.assert_array_equal(got, want[0])
        np.testing.assert_array_equal(glab, wlab[0])


def test_recipe_parameter_rows_and_resize_coefficients():
    import struct
    from spcl_amd.semi_seg.data import augment as A
    rng = random.Random(9)
    f32 = lambda i: struct.unpack("<f", struct.pack("<i", i))[0]  # noqa: E731
    for name, hw in (("acdc_pretrain", (256, 256)), ("prostate_pretrain", (224, 224)), ("acdc_label", (256, 288))):
        rec = A.RECIPES[name]
        rows = [A.draw_recipe_params(3, hw, (224, 224), rec, rng) for _ in range(500)]
        assert all(len(r) == A.RECIPE_W and r[0] == 3 and r[4] == rec["pad"] for r in rows)
        assert all(0 <= r[2] <= hw[0] + 2 * rec["pad"] - 224 and 0 <= r[3] <= hw[1] + 2 * rec["pad"] - 224 for r in rows)
        assert all(bool(r[1] & 16) == rec["crop_first"] and r[1] & 8 for r in rows)  # image rotations are BILINEAR
        lo, hi = rec["brightness"] or (1.0, 1.0)
        assert all(lo <= f32(r[5]) <= hi and lo <= f32(r[6]) <= hi for r in rows)
        if not rec["flips"]:
            assert all(r[1] & 3 == 0 for r in rows)
        # the doubles are PIL's matrix of the image the rotation acts on, the 16.16 words its FIX()
        r = rows[0]
        m = [struct.unpack("<d", struct.pack("<ii", r[14 + 2 * k], r[15 + 2 * k]))[0] for k in range(6)]
        rw, rh = ((224, 224) if rec["crop_first"] else (hw[1], hw[0]))
        assert abs(m[0] ** 2 + m[1] ** 2 - 1.0) < 1e-12 and abs(m[0] * rw / 2 + m[1] * rh / 2 + m[2] - rw / 2) < 1e-9
        assert r[8] == int(np.floor(m[0] * 65536.0 + 0.5))
    for n_in, n_out in ((256, 224), (288, 224), (224, 224), (100, 224)):
        b, kk, ks = A.resize_coeffs(n_in, n_out)
        assert len(b) == n_out and all(len(k) == ks for k in kk)
        assert all(abs(sum(k) - (1 << 22)) <= ks for k in kk)  # normalised rows, rounded per tap
        assert all(0 <= lo and lo + cnt <= n_in and 0 < cnt <= ks for lo, cnt in b)
    assert A.resize_shorter_edge((256, 320), 224) == O.resize_shorter_edge((256, 320), 224) == (224, 280)
    assert A.center_crop_row(0, (256, 241), (224, 224))[2:4] == [16, 8]  # torchvision CenterCrop: int(round(.. / 2.0))


def test_store_from_a_png_folder_with_label_maps(tmp_path):
    """``DeviceSliceStore.from_folder`` on the reference's folder layout (contrastyou/data/dataset/base.py:76-140: ``img/`` and
    ``gt/`` PNGs of the same stems, ``acdc_info.npy``): PNGs PIL itself writes -- grey levels come back as k / 255, label maps
    as uint8 class codes, scans / partitions from the stems and the info file; a missing label map is an error"""
    from PIL import Image
    from spcl_amd.semi_seg.data import ACDCSliceStore
    rs = np.random.RandomState(4)
    (tmp_path / "img").mkdir()
    (tmp_path / "gt").mkdir()
    stems, imgs, gts = [], {}, {}
    for scan, n in (("patient001_00", 7), ("patient002_01", 9)):
        for k in range(n):
            stem = f"{scan}_{k:02d}"
            a = rs.randint(0, 256, size=(40, 48)).astype(np.uint8)
            g = rs.randint(0, 4, size=(40, 48)).astype(np.uint8)
            Image.fromarray(a, "L").save(tmp_path / "img" / f"{stem}.png")
            Image.fromarray(g, "L").save(tmp_path / "gt" / f"{stem}.png")
            stems.append(stem)
            imgs[stem], gts[stem] = a, g
    np.save(tmp_path / "acdc_info.npy", {"patient001_00": 7, "patient002_01": 9})
    store = ACDCSliceStore.from_folder(str(tmp_path), device="cpu")
    assert len(store) == 16 and tuple(store.images.shape) == (16, 48, 48) and store.targets.dtype == torch.uint8
    names = store.get_memory_dictionary()["img"]
    assert names == sorted(stems) and store.get_scan_list() == ["patient001_00", "patient002_01"]
    for k, stem in enumerate(names):  # centred in the square store: 4 rows of padding above and below
        np.testing.assert_array_equal(torch.round(store.images[k, 4:44] * 255).numpy().astype(np.uint8), imgs[stem])
        np.testing.assert_array_equal(store.targets[k, 4:44].numpy(), gts[stem])
        assert float(store.images[k, :4].abs().max()) == 0.0 and int(store.targets[k, 44:].max()) == 0
    assert store.meta(0) == ("patient001_00_00", "0", "patient001_00")
    assert [store._get_partition(f) for f in names[:7]] == [O.acdc_partition(f, 7) for f in names[:7]]
    (tmp_path / "gt" / f"{names[3]}.png").unlink()
    with pytest.raises(FileNotFoundError):
        ACDCSliceStore.from_folder(str(tmp_path), device="cpu")


# ---- round 6: semi_seg/data/creator.py -- which scans train, validate and test
def _cpu_store(scans, per_scan=5, labels=True):
    from spcl_amd.semi_seg.data import ACDCSliceStore
    names = [f"{s}_{k:02d}" for s in scans for k in range(per_scan)]
    imgs = torch.arange(len(names), dtype=torch.float32)[:, None, None].expand(-1, 4, 4) / 1000.0
    return ACDCSliceStore(imgs, names, targets=(imgs * 1000).to(torch.uint8) % 4 if labels else None)


def test_creator_splits_follow_the_reference():
    """``split_dataset`` (semi_seg/data/creator.py:58-84): ``np.random.permutation`` of the SORTED scan list under
    ``fix_all_seed(seed)`` (surrounding generator states restored), cut at the running sums of the ratios;
    ``split_dataset_with_predefined_filenames`` (:38-55) and its two errors; ``extract_sub_dataset_based_on_scan_names``
    keeps the store's slice order, meta-labels and label maps; ``ScanBatchSampler`` = one batch per scan, first-seen order."""
    from spcl_amd.semi_seg.data import creator as C
    scans = ["patient100_00", "patient027_01", "patient038_01", "patient067_01", "patient003_00", "patient011_01",
             "patient050_00", "patient051_01", "patient009_01", "patient070_00"]
    store = _cpu_store(scans)
    np.random.seed(77)
    random.seed(77)
    before = (np.random.get_state()[1].copy(), random.getstate())
    for ratios, seed in (((0.5,), 1), ((0.35,), 1), ((0.2, 0.3), 5)):
        subs = C.split_dataset(store, *ratios, seed=seed)
        np.random.seed(seed)
        perm = np.random.permutation(sorted(scans)).tolist()
        np.random.seed(77)  # (what the context manager must have restored is checked below; re-arm for the next round)
        cuts = [0] + [int(len(scans) * sum(ratios[:i + 1])) for i in range(len(ratios))] + [len(scans)]
        assert [sorted(s.get_scan_list()) for s in subs] == [sorted(perm[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert sum(len(s) for s in subs) == len(store)
    sub = C.extract_sub_dataset_based_on_scan_names(store, [scans[3], scans[0]])
    keep = [i for i, s in enumerate(store.show_scan_names()) if s in (scans[0], scans[3])]
    assert torch.equal(sub.images, store.images[keep]) and torch.equal(sub.targets, store.targets[keep])
    assert sub.show_partitions() == [store.show_partitions()[i] for i in keep] and type(sub) is type(store)
    with C._seeded(3):
        np.random.rand(4), random.random(), torch.rand(2)
    assert np.array_equal(np.random.get_state()[1], before[0]) and random.getstate() == before[1]
    # the published labelled scans: 10 training scans x (k / 10)
    for k, want in ((1, ["patient100_00"]), (2, ["patient027_01", "patient100_00"]),
                    (4, ["patient027_01", "patient038_01", "patient067_01", "patient100_00"])):
        lab, unl = C.split_dataset_with_predefined_filenames(store, "acdc", labeled_ratio=float(float(k) / len(scans)))
        assert lab.get_scan_list() == sorted(want) and sorted(unl.get_scan_list()) == sorted(set(scans) - set(want))
    with pytest.raises(ValueError):
        C.split_dataset_with_predefined_filenames(store, "acdc", labeled_ratio=0.3)
    with pytest.raises(KeyError):
        C.split_dataset_with_predefined_filenames(store, "spleen", labeled_ratio=0.1)
    sampler = C.ScanBatchSampler(store)
    batches = list(sampler)
    assert len(sampler) == len(scans) == len(batches) and batches[0] == list(range(5)) and batches[-1] == list(range(45, 50))
    # get_data_loaders' refusals (:104-110,124-125) come before any device work
    C.register_dataset("acdc", lambda mode: _cpu_store(scans if mode == "train" else ["patient150_00", "patient151_01"]))
    try:
        with pytest.raises(RuntimeError):
            C.get_data_loaders({"name": "acdc", "labeled_scan_num": 11}, {}, {})
    finally:
        C._FACTORIES.pop("acdc")
    with pytest.raises(KeyError):
        C.create_dataset("mmwhsct")
