#!/usr/bin/env python
"""Debug build only (SEG_EXTRA_FLAGS=-DSEG_STAMPS): s_memtime stamps of the weight-stationary conv kernel."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from segmentation_amd import _lib as L, engine as E
lib = L.load()
lib.seg_dbg_set_stamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_stamps.restype = C.c_int
def run(hw, cin, cout, cfg, B=16):
    dt = L.SEG_BF16; dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', 3, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.1, 'biases': np.zeros(cout, np.float32)}})
    net = E.Net(store, B, dt, dev); s = torch.cuda.current_stream().cuda_stream
    p = E.Plan('pack'); net.pack(p); p.run(s)
    x = net.act(hw, hw, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype)); y = net.act(hw - 2, hw - 2, cout)
    plan = E.Plan('m'); net.conv_fwd(plan, layer, [(x, 0, 0)], hw, hw, y, cfg=cfg)
    for _ in range(3): plan.run(s)
    torch.cuda.synchronize()
    st = torch.zeros(64 * 12 * 96, dtype=torch.int64, device=dev)
    assert lib.seg_dbg_set_stamps(st.data_ptr()) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(s); e1.record(); torch.cuda.synchronize()
    lib.seg_dbg_set_stamps(None)
    a = st.cpu().numpy().reshape(64, 12, 3, 32).astype(np.float64)
    print('hw', hw, cin, '->', cout, 'cfg', cfg, 'kernel %.1f us' % (e0.elapsed_time(e1) * 1e3))
    for wg in (17,):
        t0 = a[wg, 0, 2, 0]
        ns = int((a[wg, -1, 0] > 0).sum())
        print(' wg', wg, 'start->weights landed: consumer0 %.0f loader %.0f cycles; stages %d' % (a[wg, 0, 2, 1] - t0, a[wg, -1, 2, 1] - t0, ns))
        print('  stage: loader[wait_done barrier_done] consumer0[barrier_in barrier_out compute_done]  (cycles since wg start)')
        for s_ in range(min(ns, 18)):
            print('   %2d  L %7.0f %7.0f   C %7.0f %7.0f %7.0f' % (s_, a[wg, -1, 0, s_] - t0, a[wg, -1, 1, s_] - t0, a[wg, 0, 0, s_] - t0, a[wg, 0, 1, s_] - t0, a[wg, 0, 2, 2 + s_] - t0))
import os
for c in os.environ.get('CFGS', '51,53').split(','):
    run(122, 64, 64, int(c))
