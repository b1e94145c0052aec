#!/usr/bin/env python
"""Is the eager train step host-bound?  Issues N steps without synchronising and reports the host time spent issuing them
against the time the GPU needs to drain what is left afterwards.
usage: host_bound.py [size] [batch]"""
import sys, time, torch
sys.path.insert(0, '.')
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import SyntheticDataSet

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ds = SyntheticDataSet(B, size, 4, seed=5555, n_batches=2)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=size, learning_rate=1e-4, log_dir=None, save_dir=None, use_graph=False, dtype='bf16')
for _ in range(20):
    m.train_step()
torch.cuda.synchronize()
for N in (20, 100):
    t0 = time.perf_counter()
    for _ in range(N):
        m.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('N=%3d  host issue %.4f ms/step   drain after issue %.3f ms total   wall %.4f ms/step' % (N, (t1 - t0) * 1e3 / N, (t2 - t1) * 1e3, (t2 - t0) * 1e3 / N))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    m.train_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
