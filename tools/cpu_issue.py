#!/usr/bin/env python
"""How long the HOST needs to enqueue one eager train step (queues empty: synchronise, time train_step() without waiting for the GPU)
against the GPU time of the step: if the two are close the step is launch-bound on the host side.
    python tools/cpu_issue.py [--size 256] [--batch 16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
from segmentation_amd.unet import UNetModel            # noqa: E402
ds = SyntheticDataSet(a.batch, a.size, 4)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)
for _ in range(30):
    m.train_step()
torch.cuda.synchronize()
host = []
for _ in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_step(); host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    m.train_step()
torch.cuda.synchronize()
step = (time.perf_counter() - t0) * 1e3 / 200
host.sort()
nops = sum(1 for p in [m.fwd_plan] + [s[0] for s in m.bwd_segments] + [m.upd_plan] for o in p.ops if o[1] is not None)
print('launches per step %d   host enqueue: median %.3f ms, min %.3f ms   back-to-back step %.3f ms' % (nops, host[len(host) // 2], host[0], step))
