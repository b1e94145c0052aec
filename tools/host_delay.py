#!/usr/bin/env python
"""Is the eager step sensitive to host issue time?  Wraps every C-ABI launch of the plans with a busy wait of d microseconds
and times train steps: a step time that grows with d means the GPU is (locally) starved by the host."""
import sys, time, torch
sys.path.insert(0, '.')
from segmentation_amd import engine as E
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import SyntheticDataSet

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ds = SyntheticDataSet(16, size, 4, seed=5555, n_batches=2)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=size, learning_rate=1e-4, log_dir=None, save_dir=None, use_graph=False, dtype='bf16')
DELAY = [0.0]
for plan in (m.step_plan,):
    for i, (name, fn, args) in enumerate(plan.ops):
        if fn is None:
            continue
        def wrap(f):
            def g(*a):
                if DELAY[0] > 0:
                    t = time.perf_counter() + DELAY[0] * 1e-6
                    while time.perf_counter() < t:
                        pass
                return f(*a)
            return g
        plan.ops[i] = (name, wrap(fn), args)
for d in (0.0, 1.0, 2.0, 4.0, 0.0):
    DELAY[0] = d
    for _ in range(10):
        m.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        m.train_step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('delay %.1f us/launch: host issue %.4f ms/step, step %.4f ms' % (d, (t1 - t0) * 1e3 / 50, (t2 - t0) * 1e3 / 50), flush=True)
