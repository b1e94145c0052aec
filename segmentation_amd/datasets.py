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


class _WorkerError(object):
    """An exception caught in a producer thread, carried through the queue and re-raised in the consumer."""

    def __init__(self, exc, where):
        self.exc, self.where = exc, where


def _get_checked(q, alive, what, poll=0.2):
    """q.get() that cannot hang on dead producers: polls, re-raises a producer's exception, and raises when every
    producer has exited with nothing queued."""
    while True:
        try:
            item = q.get(timeout=poll)
        except queue.Empty:
            if not alive():
                try:
                    item = q.get_nowait()         # (something may have been queued just before the last producer exited)
                except queue.Empty:
                    raise RuntimeError('%s: every producer thread has stopped and the queue is empty' % what)
            else:
                continue
        if isinstance(item, _WorkerError):
            raise RuntimeError('%s: producer thread failed in %s: %r' % (what, item.where, item.exc)) from item.exc
        return item


class ThreadedImageMaskDataSet(object):
    """Image/mask folder loader behind the reference's ImageMaskDataSet keyword set (utils/datasets.py:94-197), built on
    the producer-thread -> bounded FIFO pattern of utils/threaded_dataset.py:82-90,124-166:

      file order   one permutation per EPOCH shared by the image and the mask list (= the reference's pair of
                   string_input_producer(shuffle=True, seed=seed), utils/datasets.py:136-143)
      sample       decode, /255, joint random crop of the 3+1 channel stack (:176-190), mask -> uint8 (255 -> 1, else 0: F15)
      batching     tf.train.shuffle_batch (:166-171): decoded samples enter a pool of at most `capacity`; a batch is drawn
                   uniformly from the pool once it holds `min_holding` + batch_size samples (min_after_dequeue)
      hand-over    batches are assembled IN a ring of pre-allocated pinned host buffers (no allocation, no pin_memory() per
                   batch); get_batch() returns views of a ring slot that stay valid until the second-next get_batch()

    The defaults capacity=5000 / min_holding=1250 are the reference's shuffle_batch numbers: the first batch waits for 1250 + batch
    decodes, and the pool holds up to 5000 crops (kept as uint8: < 1 GB at 256 x 256 x 3; pass smaller numbers for a quick start).

    `ratio` is accepted and unused exactly as in the reference (its only use there, decode_jpeg(ratio=), is commented out:
    utils/datasets.py:160,164).  `threads` decode workers + one batch assembler; any exception in them (unreadable file, image
    smaller than the crop) is re-raised by get_batch() instead of hanging the training loop."""

    RING = 4

    def __init__(self, image_dir, mask_dir, image_names=None, mask_names=None, split_train_val=False, n_classes=2,
                 batch_size=96, crop_size=256, ratio=1.0, capacity=5000, image_ext='jpg', mask_ext='png', seed=5555,
                 threads=4, min_holding=1250, loader=None):
        self.image_names = sorted(glob.glob(os.path.join(image_dir, '*.' + image_ext)))
        self.mask_names = sorted(glob.glob(os.path.join(mask_dir, '*.' + mask_ext)))
        if len(self.image_names) != len(self.mask_names) or not self.image_names:
            raise Exception('image / mask lists differ or are empty')
        self.batch_size, self.crop_size, self.n_classes = batch_size, crop_size, n_classes
        self.ratio, self.capacity, self.min_holding = ratio, max(int(capacity), batch_size), max(int(min_holding), 0)
        if self.min_holding + batch_size > self.capacity:
            self.min_holding = self.capacity - batch_size
        self.has_masks, self.use_feed = True, False
        self.threads, self.seed = max(1, int(threads)), seed
        self._samples = queue.Queue(maxsize=max(2 * batch_size, 64))      # decoded samples (back-pressure on the decoders)
        self._ready = queue.Queue()                                       # ring slots holding an assembled batch
        self._free = queue.Queue()
        self._stop = threading.Event()
        self._workers = []
        self._loader = loader or self._pil_loader
        self._order_lock = threading.Lock()
        self._order_rng = np.random.default_rng(seed)
        self._order, self._pos = None, 0
        self._ring = None
        self._held = []
        self.sess = None

    @staticmethod
    def _pil_loader(path):
        from PIL import Image
        return np.asarray(Image.open(path))

    def set_tf_sess(self, sess):
        self.sess = sess

    def _next_index(self):
        with self._order_lock:
            if self._order is None or self._pos >= len(self._order):
                self._order, self._pos = self._order_rng.permutation(len(self.image_names)), 0
            i = int(self._order[self._pos]); self._pos += 1
            return i

    def _decode(self, wid):
        rng = np.random.default_rng([self.seed, 1 + wid])
        c = self.crop_size
        try:
            while not self._stop.is_set():
                i = self._next_index()
                im = self._loader(self.image_names[i])
                mk = self._loader(self.mask_names[i])
                if mk.ndim == 3:
                    mk = mk[..., 0]
                if im.ndim == 2:
                    im = np.repeat(im[..., None], 3, axis=2)
                h, w = im.shape[:2]
                if h < c or w < c or mk.shape[:2] != (h, w):
                    raise ValueError('%s is %dx%d (mask %s): cannot take a %dx%d crop' % (self.image_names[i], h, w, mk.shape[:2], c, c))
                y0, x0 = int(rng.integers(0, h - c + 1)), int(rng.integers(0, w - c + 1))
                # the pool keeps 8-bit crops as they are (a 5000-sample pool of float32 256 x 256 x 3 crops is 3.9 GB of host memory, of
                # uint8 crops under 1 GB); the /255 and the float cast happen when a batch is assembled into its ring slot -- same bits
                x = np.ascontiguousarray(im[y0:y0 + c, x0:x0 + c, :3])
                if x.dtype != np.uint8:
                    x = x.astype(np.float32) / np.float32(255.0)
                y = (mk[y0:y0 + c, x0:x0 + c].astype(np.float32) / np.float32(255.0)).astype(np.uint8)
                while not self._stop.is_set():
                    try:
                        self._samples.put((x, y), timeout=0.1)
                        break
                    except queue.Full:
                        continue
        except Exception as e:                      # noqa: surfaces in get_batch()
            self._ready.put(_WorkerError(e, 'decode worker %d' % wid))

    def _assemble(self):
        rng = np.random.default_rng([self.seed, 0])
        pool, B = [], self.batch_size
        try:
            while not self._stop.is_set():
                while len(pool) < self.min_holding + B and not self._stop.is_set():      # fill to min_after_dequeue + one batch
                    pool.append(_get_checked(self._samples, self._decoders_alive, 'ThreadedImageMaskDataSet'))
                while len(pool) < self.capacity:                                        # then take what is there, up to capacity
                    try:
                        pool.append(self._samples.get_nowait())
                    except queue.Empty:
                        break
                slot = None
                while slot is None and not self._stop.is_set():
                    try:
                        slot = self._free.get(timeout=0.1)
                    except queue.Empty:
                        continue
                if slot is None:
                    return
                img, msk = self._ring[slot]
                for b in range(B):
                    j = int(rng.integers(0, len(pool)))
                    pool[j], pool[-1] = pool[-1], pool[j]
                    x, y = pool.pop()
                    if x.dtype == np.uint8:
                        torch.div(torch.from_numpy(x).to(torch.float32), 255.0, out=img[b])
                    else:
                        img[b] = torch.from_numpy(x)
                    msk[b, ..., 0] = torch.from_numpy(y)
                self._ready.put(slot)
        except Exception as e:                      # noqa
            self._ready.put(_WorkerError(e, 'batch assembler'))

    def _decoders_alive(self):
        return any(t.is_alive() for t in self._workers[1:])

    def _producers_alive(self):
        return bool(self._workers) and self._workers[0].is_alive()

    def start(self):
        if self._workers:
            return
        c, B = self.crop_size, self.batch_size
        pin = torch.cuda.is_available()
        self._ring = []
        for i in range(self.RING):
            img = torch.empty((B, c, c, 3), dtype=torch.float32)
            msk = torch.empty((B, c, c, 1), dtype=torch.uint8)
            if pin:
                img, msk = img.pin_memory(), msk.pin_memory()
            self._ring.append((img, msk))
            self._free.put(i)
        ts = [threading.Thread(target=self._assemble, daemon=True)]
        ts += [threading.Thread(target=self._decode, args=(i,), daemon=True) for i in range(self.threads)]
        self._workers = ts
        for t in reversed(ts):
            t.start()

    def stop(self):
        """stops and JOINS the producer threads (a process must not reach interpreter shutdown with them inside torch calls)"""
        self._stop.set()
        for t in self._workers:
            if t is not threading.current_thread():
                t.join(timeout=5.0)

    def get_batch(self):
        """-> (float32 [B,c,c,3] in [0,1], uint8 [B,c,c,1]); views of a pinned ring slot, valid until the second-next call."""
        self.start()
        while len(self._held) >= 2:
            self._free.put(self._held.pop(0))
        slot = _get_checked(self._ready, self._producers_alive, 'ThreadedImageMaskDataSet')
        self._held.append(slot)
        img, msk = self._ring[slot]
        return img.numpy(), msk.numpy()

    def get_pinned_batch(self):
        """Same, as the pinned torch tensors themselves (DevicePrefetcher copies H2D straight out of them)."""
        self.start()
        while len(self._held) >= 2:
            self._free.put(self._held.pop(0))
        slot = _get_checked(self._ready, self._producers_alive, 'ThreadedImageMaskDataSet')
        self._held.append(slot)
        return self._ring[slot]


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

    def _alloc(self, img, msk, staging=True):
        """depth device slots (+ pinned staging buffers unless the dataset hands out pinned memory itself): allocated once."""
        self._slots = []
        for i in range(self.depth):
            px = torch.empty(img.shape, dtype=torch.float32).pin_memory() if staging else None
            py = torch.empty(msk.shape, dtype=torch.uint8).pin_memory() if staging else None
            dx = torch.empty(img.shape, dtype=torch.float32, device=self.device)
            dy = torch.empty(msk.shape, dtype=torch.uint8, device=self.device)
            self._slots.append({'px': px, 'py': py, 'dx': dx, 'dy': dy, 'ready': None, 'consumed': None})
            self._free.put(i)

    def _take_slot(self):
        while not self._stop.is_set():
            try:
                i = self._free.get(timeout=0.1)
            except queue.Empty:
                continue
            s = self._slots[i]
            if s['consumed'] is not None:
                s['consumed'].synchronize()          # the consumer's reads of this slot have finished
            return i, s
        return None, None

    def _produce(self, tid=0):
        try:
            torch.cuda.set_device(self.device)
            direct = hasattr(self.ds, 'get_pinned_batch')      # the loader's own pinned ring: H2D straight out of it
            if direct and tid != 0:
                return
            prev = None
            while not self._stop.is_set():
                if direct:
                    if prev is not None:
                        prev.synchronize()           # the ring slot handed out two fetches ago is recycled by the next fetch
                    img, msk = self.ds.get_pinned_batch()
                else:
                    with self._lock:
                        img, msk = self.ds.get_batch()
                    img = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img, np.float32))
                    msk = msk if isinstance(msk, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(msk, np.uint8))
                with self._alloc_lock:
                    if self._slots is None:
                        self._alloc(img, msk, staging=not direct)
                i, s = self._take_slot()
                if s is None:
                    return
                if direct:
                    hx, hy = img, msk
                else:
                    s['px'].copy_(img.reshape(s['px'].shape)); s['py'].copy_(msk.reshape(s['py'].shape))
                    hx, hy = s['px'], s['py']
                with torch.cuda.stream(self._copy):
                    s['dx'].copy_(hx.reshape(s['dx'].shape), non_blocking=True)
                    s['dy'].copy_(hy.reshape(s['dy'].shape), non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(self._copy)
                s['ready'] = ev
                prev = ev
                self._ready.put(i)
        except Exception as e:                      # noqa: surfaces in get_device_batch()
            self._ready.put(_WorkerError(e, 'DevicePrefetcher producer %d' % tid))

    def start(self):
        if self._thr is None:
            if hasattr(self.ds, 'start'):
                self.ds.start()
            self._thr = [threading.Thread(target=self._produce, args=(i,), daemon=True) for i in range(self.threads)]
            for t in self._thr:
                t.start()

    def stop(self):
        self._stop.set()
        for t in self._thr or []:
            if t is not threading.current_thread():
                t.join(timeout=5.0)
        if hasattr(self.ds, 'stop'):
            self.ds.stop()

    def get_device_batch(self):
        self.start()
        if self._last is not None:                   # the previous slot has been copied out by now: hand it back
            ev = torch.cuda.Event(); ev.record()
            self._slots[self._last]['consumed'] = ev
            self._free.put(self._last)
        i = _get_checked(self._ready, lambda: any(t.is_alive() for t in self._thr), 'DevicePrefetcher')
        s = self._slots[i]
        torch.cuda.current_stream().wait_event(s['ready'])
        self._last = i
        return s['dx'], s['dy']

    def get_batch(self):
        x, y = self.get_device_batch()
        return x.cpu().numpy(), y.cpu().numpy()
