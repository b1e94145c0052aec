#!/usr/bin/env python
"""Debug build only: where a conv_sweep_kernel launch spends its time (s_memrealtime stamps, 10 ns ticks).
    python tools/stamp_conv.py hw,cin,cout[,B[,cfg]] ..."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from segmentation_amd import _build
LIBS = os.path.join(ROOT, 'segmentation_amd', 'build', 'libseg_cstamps.so')
if 'SEG_LIB_PATH' not in os.environ:
    _build.build(verbose=False)
    d = os.path.join(ROOT, 'segmentation_amd', 'build')
    o = os.path.join(d, 'conv_sweep_stamps.o')
    subprocess.check_call([_build.HIPCC] + _build.FLAGS + ['-DSEG_STAMPS', '-c', os.path.join(_build.CSRC, 'conv_sweep.hip'), '-o', o])
    objs = [os.path.join(d, f.replace('.hip', '.o')) for f in _build.SOURCES if f != 'conv_sweep.hip'] + [o]
    subprocess.check_call([_build.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBS] + objs)
    os.environ['SEG_LIB_PATH'] = LIBS
    sys.exit(subprocess.call([sys.executable] + sys.argv))         # a child process with SEG_LIB_PATH set (never exec)
import ctypes as C, numpy as np, torch
from segmentation_amd import _lib as L, engine as E
lib = L.load()
lib.seg_dbg_set_csstamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_csstamps.restype = C.c_int


def run(hw, cin, cout, B=16, cfg=0):
    dt = L.SEG_BF16; dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', 3, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=False)
    rng = np.random.default_rng(0)
    store.set_params({'c': {'weights': rng.standard_normal(layer.wshape).astype(np.float32) * 0.05, 'biases': np.zeros(cout, np.float32)}})
    net = E.Net(store, B, dt, dev); s = torch.cuda.current_stream().cuda_stream
    pk = E.Plan('p'); net.pack(pk); pk.run(s); torch.cuda.synchronize()
    x = net.act(hw, hw, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype))
    out = net.act(hw - 2, hw - 2, cout)
    plan = E.Plan('m'); net.conv_fwd(plan, layer, [(x, 0, 0)], hw, hw, out, cfg=cfg)
    for _ in range(3): plan.run(s)
    torch.cuda.synchronize()
    st = torch.zeros(512 * 128, dtype=torch.int64, device=dev)
    assert lib.seg_dbg_set_csstamps(st.data_ptr()) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(s); e1.record(); torch.cuda.synchronize()
    lib.seg_dbg_set_csstamps(None)
    a = st.cpu().numpy().reshape(512, 2, 64).astype(np.float64)
    a = a[a[:, 0, 0] > 0]
    nch = cin // 32 if cin % 32 == 0 else (cin + 31) // 32
    print('hw %d %d->%d B %d  %s  events %.1f us; %d workgroups stamped, %d chunks per tile' % (hw, cin, cout, B, plan.kernel_name(0), e0.elapsed_time(e1) * 1e3, len(a), nch))
    t0 = a[:, 0, 0].min()
    w = a[len(a) // 2]                                   # one workgroup in the middle
    us = lambda v: (v - t0) / 100.0
    print('   workgroup %d: entry %.2f us' % (len(a) // 2, us(w[0, 0])))
    i = 1; k = 0
    while i < 64 and w[0, i] > 0 and k < 5:
        line = '   tile %d:' % k
        for c in range(nch):
            if i + 1 >= 64 or w[0, i] == 0: break
            line += '  chunk%d landed %.2f computed %.2f' % (c, us(w[0, i]), us(w[0, i + 1])); i += 2
        if i < 64 and w[0, i] > 0:
            line += '  flushed %.2f' % us(w[0, i]); i += 1
        print(line); k += 1
    ev = [us(v) for v in w[1, 1:] if v > 0][:14]
    print('   loader events (fetch issued / committed, alternating after the first two fetches):', ' '.join('%.2f' % v for v in ev))
    per_tile = []
    for r in a:
        ts = [v for v in r[0, 1:] if v > 0]
        if len(ts) >= 2 * (2 * nch + 1):
            per_tile.append((ts[2 * (2 * nch + 1) - 1] - ts[2 * nch]) / 100.0)     # flush(tile1) - flush(tile0)
    if per_tile:
        print('   steady-state time per tile (flush to flush), median over workgroups: %.2f us' % np.median(per_tile))


for arg in sys.argv[1:] or ['253,64,64', '58,256,256']:
    run(*[int(v) for v in arg.split(',')])
