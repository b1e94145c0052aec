#!/usr/bin/env python
"""Debug build only (SEG_EXTRA_FLAGS=-DSEG_ABLATE): run time of the tiled conv kernel with parts of it switched off
(patch loads, filter loads, LDS reads + MFMAs, epilogue, LDS commits), one layer shape at a time.
usage: ablate_conv.py [hw cin cout cfg]..."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from segmentation_amd import _lib as L, engine as E
lib = L.load()
HAVE = hasattr(lib, 'seg_dbg_set_ablate')        # plain builds: only the full kernel is timed
if HAVE:
    lib.seg_dbg_set_ablate.argtypes = [C.c_int]; lib.seg_dbg_set_ablate.restype = C.c_int
NAMES = [(0, 'full kernel'), (1, '- patch loads'), (2, '- filter loads'), (3, '- all global loads'), (4, '- LDS reads + MFMAs'), (8, '- epilogue'),
         (16, '- LDS commits'), (12, '- compute - epilogue (loads + commits only)'), (7, '- loads - compute (commits + epilogue)'),
         (11, '- loads - epilogue (commit + compute)'), (31, 'nothing but barriers'), (32, 'empty kernel (launch + dispatch)')]


def run(hw, cin, cout, cfg, B=16, reps=40):
    dt = L.SEG_BF16; dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', 3, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.1, 'biases': np.zeros(cout, np.float32)}})
    net = E.Net(store, B, dt, dev); s = torch.cuda.current_stream().cuda_stream
    p = E.Plan('pack'); net.pack(p); p.run(s)
    x = net.act(hw, hw, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype)); y = net.act(hw - 2, hw - 2, cout)
    plan = E.Plan('m'); net.conv_fwd(plan, layer, [(x, 0, 0)], hw, hw, y, cfg=cfg)
    print('conv 3x3 %dx%d %d->%d B=%d cfg %d' % (hw, hw, cin, cout, B, cfg))
    for bits, name in (NAMES if HAVE else NAMES[:1]):
        if HAVE:
            assert lib.seg_dbg_set_ablate(bits) == 0
        for _ in range(5): plan.run(s)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan.run(s)
        e1.record(); torch.cuda.synchronize()
        print('  %2d %-46s %7.1f us' % (bits, name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
    if HAVE:
        lib.seg_dbg_set_ablate(0)


args = [int(a) for a in sys.argv[1:]] or [125, 64, 64, 1, 252, 32, 32, 2, 59, 128, 128, 1]
for i in range(0, len(args), 4):
    run(*args[i:i + 4])
