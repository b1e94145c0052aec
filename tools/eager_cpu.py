#!/usr/bin/env python
"""CPU cost of issuing one eager train step (no graph): time around the launch loop without synchronising."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentation_amd.datasets import SyntheticDataSet
from segmentation_amd.unet import UNetModel
ds = SyntheticDataSet(16, 256, 4, seed=5555, n_batches=2)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=256, learning_rate=1e-4, log_dir=None, save_dir=None, load_snapshot=False,
              dtype='bf16', use_graph=False, seed=5555)
for _ in range(5): m.train_step()
torch.cuda.synchronize()
t = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t0))
print('issue %.0f us, issue+drain %.0f us' % (1e6 * min(a for a, b in t), 1e6 * min(b for a, b in t)))
