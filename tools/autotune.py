#!/usr/bin/env python
"""In-situ tile autotuner: for every conv / filter-gradient shape of the U-Net train step, try the other tile
configurations INSIDE the overlapped, graph-captured step and keep what makes the whole step faster (greedy, one shape
at a time, repeated passes).  Writes segmentation_amd/tuning/gfx950.json (shape signature -> cfg), which engine.Net
applies wherever a descriptor is emitted with cfg = 0.

  python tools/autotune.py [--passes 2] [--model unet|fcn8s] [--out path]
"""
import argparse, collections, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmentation_amd import _lib as L, engine as E
from segmentation_amd.datasets import SyntheticDataSet

ap = argparse.ArgumentParser()
ap.add_argument('--passes', type=int, default=2)
ap.add_argument('--model', default='unet')
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--classes', type=int, default=4)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--gain', type=float, default=0.004, help='minimum relative step-time gain to accept a change')
ap.add_argument('--only', default='', help='substring filter on the shape keys')
ap.add_argument('--out', default=os.path.join(ROOT, 'segmentation_amd', 'tuning', 'gfx950.json'))
args = ap.parse_args()

ds = SyntheticDataSet(args.batch, args.size, args.classes, seed=5555, n_batches=2)
if args.model == 'unet':
    from segmentation_amd.unet import UNetModel
    m = UNetModel(sess=None, dataset=ds, n_classes=args.classes, input_dims=args.size, learning_rate=1e-4, log_dir=None, save_dir=None,
                  load_snapshot=False, dtype='bf16', use_graph=True, seed=5555)
else:
    from segmentation_amd.fcn import FCNModel
    m = FCNModel(sess=None, dataset=ds, n_classes=args.classes, input_dims=args.size, learning_rate=1e-4, fcn_type='8s', log_dir=None,
                 save_dir=None, load_snapshot=False, dtype='bf16', use_graph=True, seed=5555)
net, lib = m.net, L.load()


def measure():
    m._graphs.clear()
    for _ in range(4):
        m.train_step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m.train_step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.steps)
    return best * 1e3


# descriptors by shape key
groups = collections.OrderedDict()
twins = {}                      # id(wgrad desc) -> its phase-2 copy
for plan in (m.fwd_plan, m.bwd_plan):
    for i, (name, fn, a) in enumerate(plan.ops):
        d = plan.meta[i].get('desc')
        if d is None:
            continue
        groups.setdefault(E.tune_key(d), []).append((plan, name, d))
# the phase-2 twin is the ctypes object referenced by the following op's byref argument
for plan in (m.fwd_plan, m.bwd_plan):
    for i, (name, fn, a) in enumerate(plan.ops):
        d = plan.meta[i].get('desc')
        if isinstance(d, L.WgradDesc) and i + 1 < len(plan.ops) and plan.ops[i + 1][0] == name + '/reduce':
            twins[id(d)] = plan.ops[i + 1][2][0]._obj


def set_cfg(key, cfg):
    for plan, name, d in groups[key]:
        d.cfg = cfg
        if isinstance(d, L.WgradDesc):
            net._wgrad_ws(d, plan, 0)                 # re-plan the K split / workspace for this tile shape
            d.phase = 1 if d.ksplit > 1 else 0
            w2 = twins.get(id(d))
            if w2 is not None:
                w2.cfg, w2.ws, w2.ws_bytes, w2.ksplit = d.cfg, d.ws, d.ws_bytes, d.ksplit
            elif d.ksplit > 1:
                raise L.SegError('no reduce op for a split that now needs one')


def candidates(key):
    # only configurations that the -m gpu kernel tests exercise for that kernel family
    if key.startswith('w:'):
        return [1, 2, 3, 4, 5, 6, 8, 9]
    if key.startswith('c:3x3s1u0'):
        return [1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 32, 34, 51, 52, 53, 54]
    return [1, 2, 3, 4, 13, 14]


trace = open(os.path.join(ROOT, 'gpurun_out', 'autotune_trace.log'), 'w') if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else sys.stderr
cur = {k: g[0][2].cfg for k, g in groups.items()}
base = measure()
print('baseline %.4f ms/step, %d shapes' % (base, len(groups)), flush=True)
best_t = base
for ps in range(args.passes):
    changed = 0
    for key in groups:
        if args.only and args.only not in key:
            continue
        keep = cur[key]
        for cand in candidates(key):
            if cand == cur[key]:
                continue
            try:
                print('    try %s cfg %d' % (key, cand), file=trace, flush=True)
                set_cfg(key, cand)
                t = measure()
            except Exception as e:
                set_cfg(key, keep)
                continue
            if t < best_t * (1 - args.gain):
                t2 = measure()                                   # confirm
                if t2 < best_t * (1 - args.gain / 2):
                    print('  %-52s cfg %2d -> %2d : %.4f -> %.4f' % (key, keep, cand, best_t, max(t, t2)), flush=True)
                    best_t, keep = max(t, t2), cand
                    changed += 1
                    continue
            set_cfg(key, keep)
        set_cfg(key, keep)
        cur[key] = keep
    print('pass %d: %d changes, %.4f ms/step' % (ps, changed, best_t), flush=True)
    if not changed:
        break
final = measure()
print('final %.4f ms/step (baseline %.4f)' % (final, base))
out = {'_about': 'tools/autotune.py on MI355X: tile cfg per shape signature (engine.tune_key), found inside the overlapped train step',
       '_baseline_ms': round(base, 4), '_tuned_ms': round(final, 4)}
if os.path.exists(args.out):
    out.update({k: v for k, v in json.load(open(args.out)).items() if not k.startswith('_')})
out.update({k: v for k, v in cur.items() if v != 0})
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(out, open(args.out, 'w'), indent=1, sort_keys=True)
print('wrote', args.out)
