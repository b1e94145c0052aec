#!/usr/bin/env python
"""Safety sweep: every explicit tile cfg on every conv flavour (3x3, 1x1, transposed-conv forward = 1x1 + scatter,
its dgrad = 2x2/s2) on small shapes, compared with the automatic choice.  Logs BEFORE each launch (a GPU fault
aborts the process: the last line names the culprit)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_amd import _lib as L, engine as E
dt = L.SEG_BF16; dev = torch.device('cuda', 0)
s = lambda: torch.cuda.current_stream().cuda_stream
log = open(sys.argv[1], 'w') if (__name__ == '__main__' and len(sys.argv) > 1) else sys.stdout
def note(*a):
    print(*a, file=log, flush=True)
CFGS = [1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 21, 22, 23, 24]
def run_plan(plan):
    plan.run(s()); torch.cuda.synchronize()
def sweep_conv(k, cin, cout, H, B, tag):
    layer = E.Layer('c', 'conv', k, cin, cout, 'VALID' if k == 3 else 'SAME', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.1, 'biases': rng.standard_normal(cout).astype(np.float32)}})
    net = E.Net(store, B, dt, dev); p = E.Plan('pack'); net.pack(p); run_plan(p)
    srcs = []
    for c in cin:
        a = net.act(H, H, c); a.t.copy_(torch.randn(a.t.shape, device=dev).to(a.t.dtype)); a.t[..., c:] = 0; srcs.append((a, 0, 0))
    Ho = H - k + 1 if k == 3 else H
    ref = None
    for cfg in [0] + CFGS:
        y = net.act(Ho, Ho, cout)
        plan = E.Plan('m')
        note('conv', tag, 'k', k, 'cin', cin, 'cout', cout, 'H', H, 'cfg', cfg)
        try:
            net.conv_fwd(plan, layer, srcs, H, H, y, cfg=cfg); run_plan(plan)
        except L.SegError as e:
            note('   rejected:', str(e)[:80]); continue
        if ref is None: ref = y.t.float().clone(); continue
        err = (y.t.float() - ref).abs().max().item()
        note('   max diff vs auto %.3g' % err)
        assert err < 0.06 * ref.abs().max().item() + 1e-3, (tag, cfg, err)
def sweep_up(cin, cout, H, B):
    layer = E.Layer('u', 'up', 2, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    rng = np.random.default_rng(1)
    store.set_params({'u': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.1, 'biases': rng.standard_normal(cout).astype(np.float32)}})
    net = E.Net(store, B, dt, dev); p = E.Plan('pack'); net.pack(p); run_plan(p)
    x = net.act(H, H, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype))
    dzu = net.act(2 * H, 2 * H, cout); dzu.t.copy_(torch.randn(dzu.t.shape, device=dev).to(dzu.t.dtype))
    ref = refd = None
    for cfg in [0] + CFGS:
        y = net.act(2 * H, 2 * H, cout); dx = net.act(H, H, cin)
        note('up fwd cin', cin, 'cout', cout, 'H', H, 'cfg', cfg)
        try:
            plan = E.Plan('m'); net.up_fwd(plan, layer, x, H, H, y, cfg=cfg); run_plan(plan)
            if ref is None: ref = y.t.float().clone()
            else: assert (y.t.float() - ref).abs().max().item() < 0.06 * ref.abs().max().item() + 1e-3, ('up', cfg)
        except L.SegError as e:
            note('   rejected:', str(e)[:80])
        note('up dgrad cin', cin, 'cout', cout, 'H', H, 'cfg', cfg)
        try:
            plan = E.Plan('m'); net.up_bwd(plan, layer, x, H, H, dzu, dx, x, cfg=cfg); net.flush_reduce(plan); run_plan(plan)
            if refd is None: refd = dx.t.float().clone()
            else: assert (dx.t.float() - refd).abs().max().item() < 0.06 * refd.abs().max().item() + 1e-3, ('updx', cfg)
        except L.SegError as e:
            note('   rejected:', str(e)[:80])
def main():
    sweep_conv(3, [64], 64, 21, 2, 'mid')
    sweep_conv(3, [256, 256], 256, 14, 4, 'deep2')
    sweep_conv(3, [512], 512, 10, 4, 'deep')
    sweep_conv(1, [96], 4, 9, 2, 'score')
    sweep_conv(1, [64], 160, 16, 2, 'wide1x1')
    sweep_up(512, 256, 8, 4)
    sweep_up(64, 32, 36, 2)
    note('sweep complete')


if __name__ == '__main__':
    main()
