#!/usr/bin/env python
"""Which stream bounds the train step?  Times eager train steps with groups of plan ops skipped (results are garbage with
anything skipped): none / the slab reductions / every filter gradient / every side-stream op.
usage: exp_crit.py [size] [batch]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from segmentation_amd import engine as E
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import SyntheticDataSet

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ds = SyntheticDataSet(B, size, 4, seed=5555, n_batches=2)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=size, learning_rate=1e-4, log_dir=None, save_dir=None, use_graph=False, dtype='bf16')
SKIP = {'pat': None}
orig = E.Plan.run


def run(self, stream, side=None, skip=(), flavor='per_layer'):
    pat = SKIP['pat']
    if pat is not None:
        skip = set(n for i, (n, fn, a) in enumerate(self.ops) if fn is not None and pat(n, self.meta[i]))
    return orig(self, stream, side, skip, flavor)


E.Plan.run = run
for label, pat in [('full step', None),
                   ('- slab reductions', lambda n, md: n.endswith('/reduce')),
                   ('- filter gradients and reductions', lambda n, md: n.endswith('/reduce') or n.endswith('/dw')),
                   ('- every side-stream op', lambda n, md: md.get('side', 0) not in (0, None)),
                   ('- side-stream ops and Adam', lambda n, md: md.get('side', 0) not in (0, None) or n.startswith('adam')),
                   ('forward only (no loss head)', lambda n, md: md.get('side', 0) not in (0, None) or n.startswith('adam') or '/d' in n or n.startswith('pool/bwd') or n.startswith('output')),
                   ('- data gradients (forward + filter gradients + Adam)', lambda n, md: ('/dx' in n or n.startswith('pool/bwd'))),
                   ('full step', None)]:
    SKIP['pat'] = pat
    for _ in range(10):
        m.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        m.train_step()
    torch.cuda.synchronize()
    print('%-40s %.4f ms/step' % (label, (time.perf_counter() - t0) * 1e3 / 50), flush=True)
