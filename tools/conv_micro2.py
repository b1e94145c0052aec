#!/usr/bin/env python
"""Stand-alone timing of the tiled convolution (forward launches; the dgrads run on the same kernel) on the U-Net's 3x3
layer shapes for a list of tile configs.
    python tools/conv_micro2.py [--size 512] [--batch 16] [--cfgs 0,1,11,15,16] [--layers conv2_2,conv3_2]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from segmentation_amd import _lib as L, engine as E      # noqa: E402
from segmentation_amd.unet import unet_sizes            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--cfgs', default='0,1,2,11,15,16,17,21')
ap.add_argument('--layers', default='')
ap.add_argument('--reps', type=int, default=30)
a = ap.parse_args()
dev = torch.device('cuda', 0)
dt = L.SEG_BF16
sh = unet_sizes(a.size)
nk = 32
shapes = [('conv1_2', [nk], nk, sh['upconv4'] + 2), ('conv2_1', [nk], 2 * nk, sh['pool1']), ('conv2_2', [2 * nk], 2 * nk, sh['conv2_1']),
          ('conv3_1', [2 * nk], 4 * nk, sh['pool2']), ('conv3_2', [4 * nk], 4 * nk, sh['conv3_1']), ('conv4_1', [4 * nk], 8 * nk, sh['pool3']),
          ('conv4_2', [8 * nk], 8 * nk, sh['conv4_1']), ('conv5_1', [8 * nk], 16 * nk, sh['pool4']), ('conv5_2', [16 * nk], 16 * nk, sh['conv5_1']),
          ('conv6_1', [8 * nk, 8 * nk], 8 * nk, sh['upconv1']), ('conv6_2', [8 * nk], 8 * nk, sh['conv6_1']),
          ('conv7_1', [4 * nk, 4 * nk], 4 * nk, sh['upconv2']), ('conv7_2', [4 * nk], 4 * nk, sh['conv7_1']),
          ('conv8_1', [2 * nk, 2 * nk], 2 * nk, sh['upconv3']), ('conv8_2', [2 * nk], 2 * nk, sh['conv8_1']),
          ('conv9_1', [nk, nk], nk, sh['upconv4']), ('conv9_2', [nk], nk, sh['conv9_1'])]
want = [s for s in a.layers.split(',') if s]
for name, segs, cout, H in shapes:
    if want and name not in want:
        continue
    layer = E.Layer('c', 'conv', 3, segs, cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=False)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.05, 'biases': np.zeros(cout, np.float32)}})
    net = E.Net(store, a.batch, dt, dev)
    pk = E.Plan('p'); net.pack(pk); pk.run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    srcs = []
    for c in segs:
        t = net.act(H, H, c); t.t.copy_(torch.randn(t.t.shape, device=dev).to(t.t.dtype)); srcs.append((t, 0, 0))
    out = net.act(H - 2, H - 2, cout)
    fl = 2 * a.batch * (H - 2) ** 2 * 9 * sum(segs) * cout
    for cfg in [int(c) for c in a.cfgs.split(',')]:
        plan = E.Plan('f')
        try:
            net.conv_fwd(plan, layer, srcs, H, H, out, cfg=cfg)
            nm, fn, args = plan.ops[0]
            sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                rc = fn(*args, sp)
                if rc:
                    L.check(rc, 'conv')
        except L.SegError as e:
            print('%-8s cfg %2d rejected: %s' % (name, cfg, str(e)[:70]))
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn(*args, sp)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        print('%-8s %4dx%-4d k%4d n%4d  cfg %2d %-48s %8.1f us %7.1f TF/s' % (name, H, H, sum(segs), cout, cfg, plan.kernel_name(0), us, fl / us / 1e6), flush=True)
