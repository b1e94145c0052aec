#!/usr/bin/env python
"""Prints the compiled (seg_plan_run) op sequence of the U-Net train step: kind, stream, name.  usage: dump_plan.py [size] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import ArrayDataSet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(1)
x = rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32); y = rng.integers(0, 4, (1, B, S, S, 1)).astype(np.uint8)
m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=4, input_dims=S, log_dir=None, save_dir=None, load_snapshot=False,
              dtype='bf16', n_kernels=32, seed=1, use_graph=False)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
cp = list(m.step_plan.__dict__['_compiled'].values())[0]
K = {0: 'launch', 1: 'fork', 2: 'waitval', 3: 'evwait'}
for k in range(cp.n):
    op = cp.ops[k]
    print('%3d %-8s s%d %s %s' % (k, K[int(op.kind)], op.stream, ('-> s%d' % op.stream2) if op.kind == 1 else ('back %d' % op.fn if op.kind == 3 else ''), cp.names[k]))
