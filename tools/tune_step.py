#!/usr/bin/env python
"""In-step tile tuning experiment: for every tiled convolution launch of the eager train step (forward and data gradients) try
each tile configuration IN the step (descriptor.cfg is read at every launch) and keep what makes the whole step faster.
    python tools/tune_step.py [--size 256] [--batch 16] [--steps 150] [--model unet]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--steps', type=int, default=150)
ap.add_argument('--cfgs', default='1,2,3,4,5,11,12,13,14,15,21,22,23,24')
ap.add_argument('--min-gain', type=float, default=0.004)
a = ap.parse_args()
from segmentation_amd import _lib as L            # noqa: E402
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
from segmentation_amd.unet import UNetModel            # noqa: E402

ds = SyntheticDataSet(a.batch, a.size, 4)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)


def timed(n):
    for _ in range(10):
        m.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m.train_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


for _ in range(30):
    m.train_step()
base = min(timed(a.steps) for _ in range(3))
print('baseline %.4f ms' % base, flush=True)
plan = m.step_plan
cfgs = [int(c) for c in a.cfgs.split(',')]
best = base
chosen = {}
for i, (name, fn, args) in enumerate(plan.ops):
    d = plan.meta[i].get('desc')
    if not isinstance(d, L.ConvDesc) or plan.meta[i].get('side', 0):
        continue
    if d.pool.ptr:
        cand = [c for c in (1, 2) if c in cfgs]
    else:
        cand = cfgs
    auto_name = plan.kernel_name(i)
    keep = 0
    res = []
    for c in cand:
        d.cfg = c
        try:
            nm = plan.kernel_name(i)
        except L.SegError:
            continue
        if nm == auto_name:
            continue
        try:
            t = timed(a.steps)
        except L.SegError as e:
            print('   %s cfg %d rejected: %s' % (name, c, str(e)[:60])); continue
        res.append((t, c, nm))
    d.cfg = 0
    if res:
        t, c, nm = min(res)
        if t < best * (1 - a.min_gain):
            d.cfg = c
            t2 = timed(a.steps)                        # confirm
            if t2 < best * (1 - a.min_gain / 2):
                best = min(t, t2); keep = c; chosen[name] = c
            else:
                d.cfg = 0
    print('%-16s auto %-46s -> %s   step %.4f ms   tried %s' % (name, auto_name, ('cfg %d' % keep) if keep else 'auto', best,
                                                               ' '.join('%d:%.3f' % (c, t) for t, c, _ in sorted(res, key=lambda r: r[1]))), flush=True)
final = min(timed(a.steps) for _ in range(3))
print('final %.4f ms (baseline %.4f)  SEG_CFG_OVERRIDE="%s"' % (final, base, ','.join('%s=%d' % kv for kv in chosen.items())))
