#!/usr/bin/env python
"""Experiment: is one B=16 step slower than two independent B=8 steps running concurrently on two streams?
(upper bound for splitting the batch across streams inside one model)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentation_amd.datasets import SyntheticDataSet
from segmentation_amd.unet import UNetModel
def mk(B):
    ds = SyntheticDataSet(B, 256, 4, seed=5555, n_batches=2)
    return UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=256, learning_rate=1e-4, log_dir=None, save_dir=None,
                     load_snapshot=False, dtype='bf16', use_graph=True, seed=5555)
def bench(models, streams, steps=40):
    for _ in range(6):
        for m, s in zip(models, streams):
            with torch.cuda.stream(s): m.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for m, s in zip(models, streams):
            with torch.cuda.stream(s): m.train_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(m.batch_size for m in models) * steps / dt
one = mk(16)
print('1 x B16           : %.0f img/s' % bench([one], [torch.cuda.Stream()]))
del one
a, b = mk(8), mk(8)
print('1 x B8            : %.0f img/s' % bench([a], [torch.cuda.Stream()]))
print('2 x B8 concurrent : %.0f img/s' % bench([a, b], [torch.cuda.Stream(), torch.cuda.Stream()]))
c, d = mk(4), mk(4)
print('4 x B{8,8,4,4}... skip')
