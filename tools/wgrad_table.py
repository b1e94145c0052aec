"""Per filter-gradient launch of the U-Net step: K splits, slab workspace, and the in-step duration of the partial-sum
kernel and of its slab reduction (Plan.run_profiled, per-layer flavour).
    python tools/wgrad_table.py [--size 512] [--batch 16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--classes', type=int, default=4)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
from segmentation_amd import _lib as L                      # noqa: E402
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
from segmentation_amd.unet import UNetModel                 # noqa: E402
ds = SyntheticDataSet(a.batch, a.size, a.classes)
m = UNetModel(sess=None, dataset=ds, n_classes=a.classes, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False,
              dtype=a.dtype, use_graph=False)
m._bind_batch(ds)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
side = m._side
tot = {}
for rep in range(5):
    for plan in [m.fwd_plan] + [s[0] if isinstance(s, tuple) else s for s in m.bwd_segments]:
        for name, kern, ms, fl, by in plan.run_profiled(m._stream(), torch, side=side):
            t = tot.setdefault(name, [kern, 0.0])
            t[1] += ms / 5
descs = {}
for plan in [s[0] if isinstance(s, tuple) else s for s in m.bwd_segments]:
    for (name, fn, args), meta in zip(plan.ops, plan.meta):
        d = meta.get('desc')
        if isinstance(d, L.WgradDesc) and not name.endswith('/reduce'):
            descs[name] = (d.ksplit, d.ws_bytes)
print('%-16s %6s %10s %9s %9s  kernel' % ('op', 'ksplit', 'slab MB', 'dw us', 'reduce us'))
s_dw = s_red = 0.0
for name, (ks, wsb) in descs.items():
    dw = tot.get(name, ['', 0.0]); rd = tot.get(name + '/reduce', ['', 0.0])
    s_dw += dw[1]; s_red += rd[1]
    print('%-16s %6d %10.2f %9.1f %9.1f  %s' % (name, ks, wsb / 1e6, dw[1] * 1e3, rd[1] * 1e3, dw[0]))
print('sum: filter gradients %.3f ms, reductions %.3f ms' % (s_dw, s_red))
