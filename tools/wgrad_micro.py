#!/usr/bin/env python
"""Stand-alone timing of the filter-gradient kernel (+ its slab reduction) on the U-Net's layer shapes, for a list of
layouts (cfg) and K splits: what a layout choice is worth before the overlapped step blurs it.
    python tools/wgrad_micro.py [--size 256] [--batch 16] [--cfgs 0,3,9,11,12] [--ks 0,64,128,256] [--layers conv2_2,conv8_2]
Prints one line per (layer, cfg, ksplit): kernel us, reduce us, TF/s of the pair."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from segmentation_amd import _lib as L, engine as E      # noqa: E402
from segmentation_amd.unet import unet_sizes            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--cfgs', default='0,3,9,11,12,14,15')
ap.add_argument('--ks', default='0')
ap.add_argument('--layers', default='')
ap.add_argument('--reps', type=int, default=30)
a = ap.parse_args()
dev = torch.device('cuda', 0)
dt = L.SEG_BF16
sh = unet_sizes(a.size)
nk = 32
# (name, cin segments, cout, input extent) -- 3x3 VALID layers of the U-Net
shapes = [('conv1_2', [nk], nk, sh['upconv4'] + 2), ('conv2_1', [nk], 2 * nk, sh['pool1']), ('conv2_2', [2 * nk], 2 * nk, sh['conv2_1']),
          ('conv3_1', [2 * nk], 4 * nk, sh['pool2']), ('conv3_2', [4 * nk], 4 * nk, sh['conv3_1']), ('conv4_1', [4 * nk], 8 * nk, sh['pool3']),
          ('conv4_2', [8 * nk], 8 * nk, sh['conv4_1']), ('conv5_1', [8 * nk], 16 * nk, sh['pool4']), ('conv5_2', [16 * nk], 16 * nk, sh['conv5_1']),
          ('conv6_1', [8 * nk, 8 * nk], 8 * nk, sh['upconv1']), ('conv6_2', [8 * nk], 8 * nk, sh['conv6_1']),
          ('conv7_1', [4 * nk, 4 * nk], 4 * nk, sh['upconv2']), ('conv7_2', [4 * nk], 4 * nk, sh['conv7_1']),
          ('conv8_1', [2 * nk, 2 * nk], 2 * nk, sh['upconv3']), ('conv8_2', [2 * nk], 2 * nk, sh['conv8_1']),
          ('conv9_1', [nk, nk], nk, sh['upconv4']), ('conv9_2', [nk], nk, sh['conv9_1'])]
want = [s for s in a.layers.split(',') if s]
s_ = lambda: torch.cuda.current_stream().cuda_stream


def timed(plan, idx, reps):
    name, fn, args = plan.ops[idx]
    import ctypes as C
    sp = C.c_void_p(s_())
    for _ in range(3):
        fn(*args, sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(*args, sp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, segs, cout, H in shapes:
    if want and name not in want:
        continue
    layer = E.Layer('c', 'conv', 3, segs, cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    net = E.Net(store, a.batch, dt, dev)
    srcs = []
    for c in segs:
        t = net.act(H, H, c); t.t.copy_(torch.randn(t.t.shape, device=dev).to(t.t.dtype)); srcs.append((t, 0, 0))
    dz = net.act(H - 2, H - 2, cout); dz.t.copy_(torch.randn(dz.t.shape, device=dev).to(dz.t.dtype))
    fl = 2 * a.batch * (H - 2) ** 2 * 9 * sum(segs) * cout
    for cfg in [int(c) for c in a.cfgs.split(',')]:
        for ks in [int(k) for k in a.ks.split(',')]:
            plan = E.Plan('w')
            try:
                net._wgrad_ws_orig = net._wgrad_ws
                if ks:
                    net._wgrad_ws = lambda w, p, ksplit=0, _o=net._wgrad_ws_orig, _k=ks: _o(w, p, _k)
                net.conv_bwd(plan, layer, srcs, H, H, dz, [None] * len(segs), wcfg=cfg)
            except L.SegError as e:
                print('%-8s cfg %2d ks %3d  rejected: %s' % (name, cfg, ks, str(e)[:60]))
                continue
            finally:
                net._wgrad_ws = net._wgrad_ws_orig
            w = plan.meta[0]['desc']
            t0 = timed(plan, 0, a.reps)
            t1 = timed(plan, 1, a.reps) if len(plan.ops) > 1 and plan.ops[1][1] is not None and w.ksplit > 1 else 0.0
            print('%-8s %4dx%-4d k%4d n%4d  cfg %2d %-34s ksplit %4d  kernel %7.1f us  reduce %6.1f us  pair %6.1f TF/s  (kernel %6.1f TF/s)'
                  % (name, H, H, sum(segs), cout, cfg, plan.kernel_name(0)[18:], w.ksplit, t0, t1, fl / (t0 + t1) / 1e6, fl / t0 / 1e6), flush=True)
