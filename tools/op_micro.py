#!/usr/bin/env python
"""Stand-alone timing of chosen launches of a model's train-step plan (back-to-back repetitions on one stream, HIP events).
    python tools/op_micro.py --model deconv --size 512 --classes 2 --ops conv1_0/im2col,bn1,conv_out"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
ap = argparse.ArgumentParser()
ap.add_argument('--model', default='unet'); ap.add_argument('--size', type=int, default=256); ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--classes', type=int, default=4); ap.add_argument('--ops', default=''); ap.add_argument('--reps', type=int, default=30)
a = ap.parse_args()
from segmentation_amd import _lib as L
from segmentation_amd.datasets import SyntheticDataSet
ds = SyntheticDataSet(a.batch, a.size, a.classes)
kw = dict(sess=None, dataset=ds, n_classes=a.classes, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)
if a.model == 'deconv':
    from segmentation_amd.deconvolution import DeconvModel as M
elif a.model == 'unet':
    from segmentation_amd.unet import UNetModel as M
else:
    from segmentation_amd.fcn import FCNModel as M
    kw['fcn_type'] = a.model[3:]
m = M(**kw)
m._bind_batch(ds)
for _ in range(2):
    m.train_step()
torch.cuda.synchronize()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
want = [s for s in a.ops.split(',') if s]
plan = m.step_plan
for i, (name, fn, args) in enumerate(plan.ops):
    if fn is None or (want and name not in want):
        continue
    d_ = plan.meta[i].get('desc')
    if isinstance(d_, L.ConvDesc):
        d_.signal = None
    for _ in range(3):
        fn(*args, sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn(*args, sp)
    e1.record(); torch.cuda.synchronize()
    print('%-26s %-40s %8.1f us' % (name, plan.kernel_name(i)[:40], e0.elapsed_time(e1) * 1e3 / a.reps))
