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
