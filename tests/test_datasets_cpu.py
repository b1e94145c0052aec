"""CPU tests of the dataset objects (duck-type of /root/reference/utils/datasets.py:121-130,151-152,194-196)."""
import numpy as np
import pytest

from segmentation_amd.datasets import SyntheticDataSet, ArrayDataSet, ThreadedImageMaskDataSet, Session, Coordinator, start_queue_runners


def test_synthetic_dataset_contract():
    ds = SyntheticDataSet(batch_size=2, crop_size=32, n_classes=4, seed=5555, n_batches=2)
    assert ds.batch_size == 2 and ds.has_masks and not ds.use_feed
    ds.set_tf_sess(None)
    img, mask = ds.get_batch()
    assert img.shape == (2, 32, 32, 3) and img.dtype == np.float32 and 0 <= img.min() and img.max() < 1
    assert mask.shape == (2, 32, 32, 1) and mask.dtype == np.uint8 and mask.max() <= 3
    img2, _ = ds.get_batch(); img3, _ = ds.get_batch()
    assert not np.array_equal(img, img2) and np.array_equal(img, img3)
    assert np.array_equal(SyntheticDataSet(2, 32, 4, seed=5555).get_batch()[0], img)


def test_threaded_image_mask_dataset(tmp_path):
    Image = pytest.importorskip('PIL.Image')
    fd, md = tmp_path / 'f', tmp_path / 'm'
    fd.mkdir(); md.mkdir()
    rng = np.random.default_rng(0)
    for i in range(4):
        im = rng.integers(0, 256, (48, 56, 3)).astype(np.uint8)
        mk = np.zeros((48, 56), np.uint8); mk[10:30, 5 + i:40] = 255; mk[0, 0] = 128      # 128 -> 0 (binary quirk F15)
        Image.fromarray(im).save(fd / ('%02d.png' % i)); Image.fromarray(mk).save(md / ('%02d.png' % i))
    ds = ThreadedImageMaskDataSet(str(fd), str(md), batch_size=3, crop_size=32, image_ext='png', mask_ext='png', threads=2, capacity=2)
    coord = Coordinator(); threads = start_queue_runners(coord=coord, datasets=[ds])
    try:
        for _ in range(3):
            img, mask = ds.get_batch()
            assert img.shape == (3, 32, 32, 3) and img.dtype == np.float32 and img.max() <= 1.0
            assert mask.shape == (3, 32, 32, 1) and mask.dtype == np.uint8 and set(np.unique(mask)) <= {0, 1}
    finally:
        coord.request_stop(); coord.join(threads)
    with Session() as s:
        with pytest.raises(Exception):
            s.run(None)


def test_array_dataset_cycles():
    x = np.zeros((2, 1, 8, 8, 3), np.float32); x[1] = 1
    ds = ArrayDataSet(x, np.zeros((2, 1, 8, 8, 1), np.uint8))
    assert ds.get_batch()[0].max() == 0 and ds.get_batch()[0].min() == 1 and ds.get_batch()[0].max() == 0


def _write_folder(tmp_path, n=6, hw=(48, 56)):
    Image = pytest.importorskip('PIL.Image')
    fd, md = tmp_path / 'f', tmp_path / 'm'
    fd.mkdir(); md.mkdir()
    rng = np.random.default_rng(1)
    for i in range(n):
        im = np.full(hw + (3,), i * 10, np.uint8)              # the image index is recoverable from any pixel
        im[..., 1] = rng.integers(0, 256, hw)
        mk = np.zeros(hw, np.uint8); mk[:, : hw[1] // 2] = 255
        Image.fromarray(im).save(fd / ('%02d.png' % i)); Image.fromarray(mk).save(md / ('%02d.png' % i))
    return str(fd), str(md)


def test_threaded_loader_epoch_shuffle_alignment_and_ring(tmp_path):
    """One permutation per epoch shared by images and masks (utils/datasets.py:136-143): with min_holding 0 and a pool of one
    batch every epoch of 6 files shows each file exactly once; batches live in the pre-allocated ring (no new buffers)."""
    fd, md = _write_folder(tmp_path)
    ds = ThreadedImageMaskDataSet(fd, md, batch_size=3, crop_size=32, image_ext='png', threads=1, capacity=3, min_holding=0, ratio=0.5)
    seen, ptrs = [], set()
    try:
        for _ in range(8):
            img, mask = ds.get_batch()
            ptrs.add(img.__array_interface__['data'][0])
            seen += [int(round(float(v) * 255 / 10)) for v in img[:, 0, 0, 0]]
            assert set(np.unique(mask)) <= {0, 1}
    finally:
        ds.stop()
    assert len(ptrs) <= ThreadedImageMaskDataSet.RING
    for e in range(4):
        assert sorted(seen[6 * e:6 * e + 6]) == list(range(6)), seen


def test_threaded_loader_min_holding_shuffles_across_files(tmp_path):
    fd, md = _write_folder(tmp_path, n=6)
    ds = ThreadedImageMaskDataSet(fd, md, batch_size=2, crop_size=32, image_ext='png', threads=2, capacity=12, min_holding=8)
    try:
        img, _ = ds.get_batch()
        assert img.shape == (2, 32, 32, 3)
    finally:
        ds.stop()


def test_threaded_loader_surfaces_worker_errors(tmp_path):
    """An image smaller than the crop (tf.random_crop would raise) must fail get_batch(), not hang it."""
    fd, md = _write_folder(tmp_path, n=3, hw=(24, 24))
    ds = ThreadedImageMaskDataSet(fd, md, batch_size=2, crop_size=32, image_ext='png', threads=2, capacity=4, min_holding=0)
    with pytest.raises(RuntimeError, match='cannot take a 32x32 crop'):
        ds.get_batch()
    ds.stop()
