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


class DevicePrefetcher(object):
    """Host dataset -> HBM pipeline (the north-star's "pinned host buffers drained by hipMemcpyAsync"):
    a producer thread pulls batches from any host dataset (get_batch -> ndarray), writes them into a ring of pinned
    staging buffers and enqueues the H2D copies on a dedicated copy stream into a ring of device slots, recording an
    event per slot.  get_device_batch() makes the *compute* stream wait for the slot's event and hands out the device
    tensors, so the copy of batch k+1 overlaps the train step of batch k.  Slot reuse is safe as long as the consumer
    has copied the tensors out (BaseModel does a D2D copy into its static graph inputs) before `depth` more batches are
    requested; a per-slot "consumed" event enforces it."""

    def __init__(self, dataset, depth=3, device=None, threads=1):
        self.ds = dataset
        self.threads = threads
        self._lock = threading.Lock()
        self.batch_size = dataset.batch_size
        self.has_masks, self.use_feed = True, False
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.depth = depth
        self._copy = torch.cuda.Stream(self.device)
        self._free = queue.Queue()
        self._ready = queue.Queue()
        self._slots = None
        self._stop = threading.Event()
        self._thr = None
        self._last = None
        self._alloc_lock = threading.Lock()

    def set_tf_sess(self, sess):
        if hasattr(self.ds, 'set_tf_sess'):
            self.ds.set_tf_sess(sess)

    def _alloc(self, img, msk):
        self._slots = []
        for i in range(self.depth):
            px = torch.empty(img.shape, dtype=torch.float32).pin_memory()
            py = torch.empty(msk.shape, dtype=torch.uint8).pin_memory()
            dx = torch.empty(img.shape, dtype=torch.float32, device=self.device)
            dy = torch.empty(msk.shape, dtype=torch.uint8, device=self.device)
            self._slots.append({'px': px, 'py': py, 'dx': dx, 'dy': dy, 'ready': None, 'consumed': None})
            self._free.put(i)

    def _produce(self):
        torch.cuda.set_device(self.device)
        while not self._stop.is_set():
            with self._lock:
                img, msk = self.ds.get_batch()
            img = img.numpy() if isinstance(img, torch.Tensor) else np.asarray(img, np.float32)
            msk = msk.numpy() if isinstance(msk, torch.Tensor) else np.asarray(msk, np.uint8)
            with self._alloc_lock:
                if self._slots is None:
                    self._alloc(img, msk)
            while not self._stop.is_set():
                try:
                    i = self._free.get(timeout=0.1)
                    break
                except queue.Empty:
                    continue
            else:
                return
            s = self._slots[i]
            if s['consumed'] is not None:
                s['consumed'].synchronize()          # the consumer's D2D copy out of this slot has finished
            s['px'].copy_(torch.from_numpy(np.ascontiguousarray(img)))
            s['py'].copy_(torch.from_numpy(np.ascontiguousarray(msk)))
            with torch.cuda.stream(self._copy):
                s['dx'].copy_(s['px'], non_blocking=True)
                s['dy'].copy_(s['py'], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self._copy)
            s['ready'] = ev
            self._ready.put(i)

    def start(self):
        if self._thr is None:
            if hasattr(self.ds, 'start'):
                self.ds.start()
            self._thr = [threading.Thread(target=self._produce, daemon=True) for _ in range(self.threads)]
            for t in self._thr:
                t.start()

    def stop(self):
        self._stop.set()
        if hasattr(self.ds, 'stop'):
            self.ds.stop()

    def get_device_batch(self):
        self.start()
        if self._last is not None:                   # the previous slot has been copied out by now: hand it back
            ev = torch.cuda.Event(); ev.record()
            self._slots[self._last]['consumed'] = ev
            self._free.put(self._last)
        i = self._ready.get()
        s = self._slots[i]
        torch.cuda.current_stream().wait_event(s['ready'])
        self._last = i
        return s['dx'], s['dy']

    def get_batch(self):
        x, y = self.get_device_batch()
        return x.cpu().numpy(), y.cpu().numpy()
