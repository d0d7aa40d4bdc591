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
    views = PretrainViews(store.images, (224, 224))
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
