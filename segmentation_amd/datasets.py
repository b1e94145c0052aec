"""Dataset objects with the duck-type the reference models read
(/root/reference/utils/datasets.py:121-130,151-152,173-174,194-196): attributes batch_size, use_feed,
has_masks, image_op, mask_op; methods set_tf_sess(sess), get_batch().

SyntheticDataSet keeps its tensors resident in HBM (bench / parity runs: images U[0,1) float32,
labels randint(0, n_classes) uint8, numpy seed 5555 = the reference's seed, utils/datasets.py:108).
ThreadedImageMaskDataSet re-creates the reference's producer-thread -> bounded-queue pattern
(utils/threaded_dataset.py:82-90,124-166) with a ring of pinned host buffers.
"""
import glob
import os
import queue
import threading

import numpy as np
import torch


class Session(object):
    """Stand-in for tf.Session so that driver scripts keep the shape of examples/example_fcn.py:53."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, *a, **k):
        raise Exception('there is no TensorFlow session: call model.train_step() / model.infer()')


class Coordinator(object):
    """No-op replacement of tf.train.Coordinator (examples/example_fcn.py:92-93,142-143)."""

    def request_stop(self):
        pass

    def join(self, threads=None):
        for t in threads or []:
            if hasattr(t, 'stop'):
                t.stop()


def start_queue_runners(coord=None, datasets=()):
    for d in datasets:
        if hasattr(d, 'start'):
            d.start()
    return list(datasets)


class SyntheticDataSet(object):
    def __init__(self, batch_size=16, crop_size=256, n_classes=2, input_channel=3, seed=5555, device=None, n_batches=1):
        self.batch_size, self.crop_size, self.n_classes = batch_size, crop_size, n_classes
        self.has_masks, self.use_feed = True, False
        rng = np.random.default_rng(seed)
        self.images = rng.uniform(0.0, 1.0, (n_batches, batch_size, crop_size, crop_size, input_channel)).astype(np.float32)
        self.masks = rng.integers(0, n_classes, (n_batches, batch_size, crop_size, crop_size, 1)).astype(np.uint8)
        self.image_op, self.mask_op = self.images[0], self.masks[0]
        self._i = 0
        self._dev = None
        self.device = device
        self.sess = None

    def set_tf_sess(self, sess):
        self.sess = sess

    def get_batch(self):
        i = self._i % self.images.shape[0]
        self._i += 1
        return self.images[i], self.masks[i]

    def get_device_batch(self):
        if self._dev is None:
            dev = self.device if self.device is not None else torch.device('cuda', torch.cuda.current_device())
            self._dev = (torch.from_numpy(self.images).to(dev), torch.from_numpy(self.masks).to(dev))
        i = self._i % self.images.shape[0]
        self._i += 1
        return self._dev[0][i], self._dev[1][i]


class ArrayDataSet(object):
    """Feeds fixed numpy arrays (tests): images [N,B,H,W,C] float32, masks [N,B,H,W,1] uint8."""

    def __init__(self, images, masks):
        self.images, self.masks = np.asarray(images, np.float32), np.asarray(masks, np.uint8)
        self.batch_size = self.images.shape[1]
        self.crop_size = self.images.shape[2]
        self.has_masks, self.use_feed = True, False
        self._i = 0

    def set_tf_sess(self, sess):
        self.sess = sess

    def get_batch(self):
        i = self._i % self.images.shape[0]
        self._i += 1
        return self.images[i], self.masks[i]


class ThreadedImageMaskDataSet(object):
    """Image/mask folder loader: aligned shuffle (one permutation for both lists = the reference's same-seed
    string_input_producer pair, utils/datasets.py:136-143), /255, joint random crop of the 3+1 channel stack
    (:176-190), binary masks (255 -> 1, else 0; F15), batches assembled by producer threads into a bounded
    queue of pinned host buffers (back-pressure = queue full)."""

    def __init__(self, image_dir, mask_dir, n_classes=2, batch_size=96, crop_size=256, ratio=1.0, capacity=8,
                 image_ext='jpg', mask_ext='png', seed=5555, threads=4, min_holding=0, loader=None):
        self.image_names = sorted(glob.glob(os.path.join(image_dir, '*.' + image_ext)))
        self.mask_names = sorted(glob.glob(os.path.join(mask_dir, '*.' + mask_ext)))
        if len(self.image_names) != len(self.mask_names) or not self.image_names:
            raise Exception('image / mask lists differ or are empty')
        self.batch_size, self.crop_size, self.n_classes = batch_size, crop_size, n_classes
        self.has_masks, self.use_feed = True, False
        self.threads, self.seed = threads, seed
        self.q = queue.Queue(maxsize=capacity)
        self._stop = threading.Event()
        self._workers = []
        self._loader = loader or self._pil_loader
        self.sess = None

    @staticmethod
    def _pil_loader(path):
        from PIL import Image
        return np.asarray(Image.open(path))

    def set_tf_sess(self, sess):
        self.sess = sess

    def _produce(self, wid):
        rng = np.random.default_rng(self.seed + wid)
        n, c = len(self.image_names), self.crop_size
        pin = torch.cuda.is_available()
        while not self._stop.is_set():
            img = torch.empty((self.batch_size, c, c, 3), dtype=torch.float32)
            msk = torch.empty((self.batch_size, c, c, 1), dtype=torch.uint8)
            if pin:
                img, msk = img.pin_memory(), msk.pin_memory()
            for b in range(self.batch_size):
                i = int(rng.integers(0, n))
                im = self._loader(self.image_names[i]).astype(np.float32) / 255.0
                mk = self._loader(self.mask_names[i]).astype(np.float32) / 255.0
                if mk.ndim == 3:
                    mk = mk[..., 0]
                h, w = im.shape[:2]
                y0, x0 = int(rng.integers(0, h - c + 1)), int(rng.integers(0, w - c + 1))
                img[b] = torch.from_numpy(np.ascontiguousarray(im[y0:y0 + c, x0:x0 + c, :3]))
                msk[b, ..., 0] = torch.from_numpy(np.ascontiguousarray(mk[y0:y0 + c, x0:x0 + c]).astype(np.uint8))
            while not self._stop.is_set():
                try:
                    self.q.put((img, msk), timeout=0.1)
                    break
                except queue.Full:
                    continue

    def start(self):
        if self._workers:
            return
        for i in range(self.threads):
            t = threading.Thread(target=self._produce, args=(i,), daemon=True)
            t.start()
            self._workers.append(t)

    def stop(self):
        self._stop.set()

    def get_batch(self):
        self.start()
        img, msk = self.q.get()
        return img.numpy(), msk.numpy()
