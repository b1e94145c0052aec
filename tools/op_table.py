"""Every launch of the U-Net / FCN train step in program order with its stream and its in-step duration
(Plan.run_profiled: events on the stream the kernel runs on, the side streams overlapping as in the timed step).
    python tools/op_table.py [--model unet|fcn8s] [--size 256] [--batch 16] [--classes 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='unet')
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--classes', type=int, default=4)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
ds = SyntheticDataSet(a.batch, a.size, a.classes)
kw = dict(sess=None, dataset=ds, n_classes=a.classes, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False,
          dtype=a.dtype, use_graph=False)
if a.model == 'unet':
    from segmentation_amd.unet import UNetModel
    m = UNetModel(**kw)
elif a.model == 'deconv':
    from segmentation_amd.deconvolution import DeconvModel
    m = DeconvModel(**kw)
else:
    from segmentation_amd.fcn import FCNModel
    m = FCNModel(fcn_type=a.model[3:], **kw)
m._bind_batch(ds)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
plans = [m.fwd_plan] + [s[0] for s in m.bwd_segments] + [m.upd_plan]
R = 5
rows = None
for rep in range(R):
    cur = []
    for plan in plans:
        tags = {}
        for (name, fn, args), meta in zip(plan.ops, plan.meta):
            tags[name] = meta.get('side', 0)
        for name, kern, ms, fl, by in plan.run_profiled(m._stream(), torch, side=m._side):
            cur.append([plan.name, name, kern, tags.get(name, 0), ms, fl, by])
    if rows is None:
        rows = cur
    else:
        for r, c in zip(rows, cur):
            r[4] += c[4]
tot = {}
print('%-6s %-26s %-5s %9s %9s %9s  kernel' % ('plan', 'op', 'strm', 'us', 'TF/s', 'GB/s'))
for pl, name, kern, tag, ms, fl, by in rows:
    us = ms / R * 1e3
    tot[tag] = tot.get(tag, 0.0) + us
    print('%-6s %-26s %-5s %9.1f %9s %9s  %s' % (pl, name, tag, us, '%.0f' % (fl / us / 1e6) if fl else '', '%.0f' % (by / us / 1e3) if by else '', kern))
print('sum per stream tag [us]:', {str(k): round(v, 1) for k, v in tot.items()})
