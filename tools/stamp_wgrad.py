#!/usr/bin/env python
"""Debug build only (SEG_EXTRA_FLAGS=-DSEG_STAMPS): s_memtime stamps of the filter-gradient kernel's tile walk."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from segmentation_amd import _lib as L, engine as E
lib = L.load()
lib.seg_dbg_set_wstamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_wstamps.restype = C.c_int
def run(hw, cin, cout, B=16):
    dt = L.SEG_BF16; dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', 3, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    net = E.Net(store, B, dt, dev); s = torch.cuda.current_stream().cuda_stream
    x = net.act(hw, hw, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype))
    dz = net.act(hw - 2, hw - 2, cout); dz.t.copy_(torch.randn(dz.t.shape, device=dev).to(dz.t.dtype))
    plan = E.Plan('m'); net.conv_bwd(plan, layer, [(x, 0, 0)], hw, hw, dz, [None]); net.flush_reduce(plan)
    for _ in range(3): plan.run(s)
    torch.cuda.synchronize()
    st = torch.zeros(64 * 128, dtype=torch.int64, device=dev)
    assert lib.seg_dbg_set_wstamps(st.data_ptr()) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(s); e1.record(); torch.cuda.synchronize()
    lib.seg_dbg_set_wstamps(None)
    a = st.cpu().numpy().reshape(64, 4, 32).astype(np.float64)
    print('hw', hw, cin, '->', cout, ' wgrad+reduce %.1f us' % (e0.elapsed_time(e1) * 1e3), [o[0] for o in plan.ops])
    for wg in (0, 9):
        t0 = a[wg, 3, 0]
        n = int((a[wg, 0] > 0).sum())
        print(' split', wg, 'tiles', n, 'lifetime %.0f' % (a[wg, 3, 1] - t0))
        print('   tile: loads_landed  committed  computed   (ticks since start)')
        for i in range(min(n, 6)):
            print('   %2d  %7.0f %7.0f %7.0f' % (i, a[wg, 0, i] - t0, a[wg, 1, i] - t0, a[wg, 2, i] - t0))
        m = min(n, 32)
        if m > 3:
            wait = np.mean([a[wg, 0, i + 1] - a[wg, 2, i] for i in range(1, m - 1)])
            commit = np.mean([a[wg, 1, i] - a[wg, 0, i] for i in range(1, m - 1)])
            comp = np.mean([a[wg, 2, i] - a[wg, 1, i] for i in range(1, m - 1)])
            print('   mean per tile: wait for loads %.0f, barrier+commit+barrier %.0f, prefetch issue + MFMA loop %.0f ticks' % (wait, commit, comp))
import sys as _s
for _a in ([tuple(int(v) for v in x.split(",")) for x in _s.argv[1:]] or [(122, 64, 64), (59, 128, 128)]):
    run(*_a)
