#!/usr/bin/env python
"""Single-layer micro-benchmark for kernel tuning (not part of the product path):
   python tools/conv_micro.py --hw 125 --cin 64 --cout 64 --batch 16 --kind fwd --cfg 0 --iters 50
Prints the HIP-event average of one launch and the achieved TFLOP/s; meant to be run alone under
rocprofv3 --pmc for counter collection."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from segmentation_amd import _lib as L          # noqa: E402
from segmentation_amd import engine as E        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hw', type=int, default=125)
    ap.add_argument('--cin', type=int, default=64)
    ap.add_argument('--cout', type=int, default=64)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--k', type=int, default=3)
    ap.add_argument('--kind', default='fwd', choices=['fwd', 'dgrad', 'wgrad'])
    ap.add_argument('--cfg', type=int, default=0)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--dtype', default='bf16')
    a = ap.parse_args()
    dt = L.SEG_BF16 if a.dtype == 'bf16' else L.SEG_F32
    dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', a.k, [a.cin], a.cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.1,
                            'biases': np.zeros(a.cout, np.float32)}})
    net = E.Net(store, a.batch, dt, dev)
    s = torch.cuda.current_stream().cuda_stream
    p = E.Plan('pack'); net.pack(p); p.run(s)
    H = a.hw
    Ho = H - a.k + 1
    x = net.act(H, H, a.cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype))
    y = net.act(Ho, Ho, a.cout)
    dz = net.act(Ho, Ho, a.cout); dz.t.copy_(torch.randn(dz.t.shape, device=dev).to(dz.t.dtype))
    dx = net.act(H, H, a.cin)
    plan = E.Plan('m')
    if a.kind == 'fwd':
        net.conv_fwd(plan, layer, [(x, 0, 0)], H, H, y, cfg=a.cfg)
    else:
        full = E.Plan('b')
        net.conv_bwd(full, layer, [(x, 0, 0)], H, H, dz, [(dx, (0, 0), x, (0, 0))], cfg=a.cfg, wcfg=a.cfg if a.kind == 'wgrad' else 0)
        want = '/dw' if a.kind == 'wgrad' else '/dx'
        for i, (name, fn, args) in enumerate(full.ops):
            if want in name:
                plan.ops.append(full.ops[i]); plan.meta.append(full.meta[i])
        plan.keep = full.keep
    flops = 2.0 * a.batch * Ho * Ho * a.k * a.k * a.cin * a.cout
    for _ in range(5):
        plan.run(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        plan.run(s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print('%s %s hw=%d cin=%d cout=%d B=%d cfg=%d: %.2f us  %.1f TFLOP/s  [%s]' %
          (a.kind, a.dtype, H, a.cin, a.cout, a.batch, a.cfg, us, flops / us / 1e6, plan.kernel_name(0)))


if __name__ == '__main__':
    main()
