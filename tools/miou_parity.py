#!/usr/bin/env python
"""mIoU parity of the bf16 path against the f32 (exact-MFMA) path on a LEARNABLE synthetic task (BASELINE metric: "...; mIoU parity").
U-Net 256x256x3, 4 classes, batch 16 (config C2): images of random discs / rectangles on a noisy background, the label of a pixel
is the class of the shape covering it.  Same data, same initial weights, same number of steps in both dtypes; evaluation on held-
out batches through infer().  Prints one JSON line (copied to profiles/r02_miou_parity.json).
usage: miou_parity.py [steps] [n_train_batches]"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import ArrayDataSet

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B, S, NC = 16, 256, 4


def make(n, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, B, S, S, 3), np.float32); y = np.zeros((n, B, S, S, 1), np.uint8)
    yy, xx = np.mgrid[0:S, 0:S]
    col = np.array([[0.2, 0.2, 0.2], [0.9, 0.3, 0.2], [0.2, 0.8, 0.3], [0.3, 0.3, 0.9]], np.float32)
    for i in range(n):
        for b in range(B):
            lab = np.zeros((S, S), np.uint8)
            for _ in range(6):
                c = int(rng.integers(1, NC)); cy, cx, r = rng.integers(40, S - 40), rng.integers(40, S - 40), rng.integers(12, 40)
                if rng.random() < 0.5:
                    m = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
                else:
                    m = (abs(yy - cy) < r) & (abs(xx - cx) < r * 0.7)
                lab[m] = c
            img = col[lab] + rng.normal(0, 0.15, (S, S, 3)).astype(np.float32)
            x[i, b] = np.clip(img, 0, 1); y[i, b, :, :, 0] = lab
    return x, y


def miou(pred, label):
    ious = []
    for c in range(NC):
        p, l = pred == c, label == c
        u = (p | l).sum()
        if u:
            ious.append((p & l).sum() / u)
    return float(np.mean(ious))


xtr, ytr = make(NB, 1); xte, yte = make(4, 2)
out = {'workload': 'U-Net 256x256x3 4-class batch=16, %d train steps on %d synthetic shape batches, lr 1e-3' % (STEPS, NB), 'steps': STEPS}
for dt in ('f32', 'bf16'):
    m = UNetModel(sess=None, dataset=ArrayDataSet(xtr, ytr), n_classes=NC, input_dims=S, learning_rate=1e-3, log_dir=None, save_dir=None,
                  load_snapshot=False, dtype=dt, seed=5555)
    t0 = time.time(); losses = []
    for k in range(STEPS):
        m.train_step()
        if (k + 1) % 50 == 0:
            losses.append(round(m.last_loss(), 4))
    torch.cuda.synchronize()
    oh = m.out_hw[0]; o = (S - oh) // 2
    ious = []
    for i in range(xte.shape[0]):
        sig, arg = m.infer(xte[i])
        ious.append(miou(arg[..., 0].astype(np.int64), yte[i, :, o:o + oh, o:o + oh, 0]))
    out[dt] = {'loss_every_50_steps': losses, 'miou_heldout': round(float(np.mean(ious)), 4), 'train_seconds': round(time.time() - t0, 2)}
out['miou_delta_bf16_minus_f32'] = round(out['bf16']['miou_heldout'] - out['f32']['miou_heldout'], 4)
print(json.dumps(out))
