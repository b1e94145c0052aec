#!/usr/bin/env python
"""How long does the critical stream of a train step idle at the join in front of the optimizer update (waiting for the last
filter gradients)?  Two timing events on the main stream around the backward plan's final join (Plan.probe), per-launch walk
(SEG_PLAN_C=0 is set here), no tracer: usage tail_probe.py [size] [batch] [steps]"""
import os, sys
os.environ['SEG_PLAN_C'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import ArrayDataSet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(1)
x = rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32); y = rng.integers(0, 4, (1, B, S, S, 1)).astype(np.uint8)
m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=4, input_dims=S, log_dir=None, save_dir=None, load_snapshot=False,
              dtype='bf16', n_kernels=32, seed=1, use_graph=False)
for _ in range(10):
    m.train_step()
torch.cuda.synchronize()
m.step_plan.probe = []
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(N):
    m.train_step()
t1.record(); torch.cuda.synchronize()
pr = m.step_plan.probe
waits = sorted(pr[i].elapsed_time(pr[i + 1]) * 1e3 for i in range(0, len(pr), 2))
print('size %d batch %d: step %.1f us (with probe events); idle at the final join: median %.1f us, min %.1f, max %.1f (%d joins/step)' % (
    S, B, t0.elapsed_time(t1) * 1e3 / N, waits[len(waits) // 2], waits[0], waits[-1], len(pr) // 2 // N))
