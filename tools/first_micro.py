#!/usr/bin/env python
"""Stand-alone timing of the bandwidth-bound ends of the U-Net step: conv1_1 (+pool1) forward, conv1_1's filter gradient with
the fused pool1 backward, the max-pool backward of pool2/pool3, Adam and the weight re-pack.  Each launch is timed over `reps`
back-to-back repetitions on one stream (HIP events); GB/s = algorithmic bytes / time.
    python tools/first_micro.py [--size 256] [--batch 16] [--reps 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--reps', type=int, default=50)
ap.add_argument('--only', default='')
a = ap.parse_args()
from segmentation_amd import engine as E            # noqa: E402
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
from segmentation_amd.unet import UNetModel            # noqa: E402

ds = SyntheticDataSet(a.batch, a.size, 4)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)
m._bind_batch(ds)
for _ in range(2):
    m.train_step()
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
want = [s for s in a.only.split(',') if s]
plan = m.step_plan
import ctypes as C            # noqa: E402
for i, (name, fn, args) in enumerate(plan.ops):
    md = plan.meta[i]
    pick = (name.startswith('conv1_1') or name.startswith('pool/bwd') or name.startswith('adam') or name.startswith('pack') or
            name in ('conv1_2', 'conv2_1', 'conv9_2', 'conv1_2/dx0', 'conv2_1/dx0', 'step_begin') or name.startswith('output'))
    if fn is None or not pick or (want and not any(name.startswith(w) for w in want)):
        continue
    d_ = md.get('desc')
    if isinstance(d_, E.L.ConvDesc):
        d_.signal = None
    sp = C.c_void_p(stream)
    for _ in range(3):
        fn(*args, sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn(*args, sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    by = md.get('bytes', 0)
    print('%-22s %-34s %8.1f us %8.1f MB %8.0f GB/s' % (name, plan.kernel_name(i)[:34], us, by / 1e6, by / us / 1e3 if by else 0))
