#!/usr/bin/env python
"""Debug build only: where a wgrad_sweep_kernel launch spends its time.  Builds segmentation_amd/build/libseg_stamps.so
(-DSEG_STAMPS) next to the product library, runs one layer shape and prints the s_memrealtime stamps (10 ns ticks).
    python tools/stamp_sweep.py hw,cin,cout[,B[,cfg[,ksplit]]] ..."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from segmentation_amd import _build
LIBS = os.path.join(ROOT, 'segmentation_amd', 'build', 'libseg_stamps.so')
if 'SEG_LIB_PATH' not in os.environ:
    _build.build(verbose=False)
    d = os.path.join(ROOT, 'segmentation_amd', 'build')
    o = os.path.join(d, 'wgrad_sweep_stamps.o')
    subprocess.check_call([_build.HIPCC] + _build.FLAGS + ['-DSEG_STAMPS', '-c', os.path.join(_build.CSRC, 'wgrad_sweep.hip'), '-o', o])
    objs = [os.path.join(d, f.replace('.hip', '.o')) for f in _build.SOURCES if f != 'wgrad_sweep.hip'] + [o]
    subprocess.check_call([_build.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBS] + objs)
    os.environ['SEG_LIB_PATH'] = LIBS
    sys.exit(subprocess.call([sys.executable] + sys.argv))         # a child process with SEG_LIB_PATH set (never exec)
import ctypes as C, numpy as np, torch
from segmentation_amd import _lib as L, engine as E
lib = L.load()
lib.seg_dbg_set_swstamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_swstamps.restype = C.c_int


def run(hw, cin, cout, B=16, cfg=0, ks=0):
    dt = L.SEG_BF16; dev = torch.device('cuda', 0)
    layer = E.Layer('c', 'conv', 3, [cin], cout, 'VALID', True)
    store = E.ParamStore([layer], dt, dev, training=True)
    net = E.Net(store, B, dt, dev); s = torch.cuda.current_stream().cuda_stream
    x = net.act(hw, hw, cin); x.t.copy_(torch.randn(x.t.shape, device=dev).to(x.t.dtype))
    dz = net.act(hw - 2, hw - 2, cout); dz.t.copy_(torch.randn(dz.t.shape, device=dev).to(dz.t.dtype))
    plan = E.Plan('m'); net.conv_bwd(plan, layer, [(x, 0, 0)], hw, hw, dz, [None], wcfg=cfg, ksplit=ks)
    for _ in range(3): plan.run(s)
    torch.cuda.synchronize()
    st = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
    assert lib.seg_dbg_set_swstamps(st.data_ptr()) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(s); e1.record(); torch.cuda.synchronize()
    lib.seg_dbg_set_swstamps(None)
    a = st.cpu().numpy().reshape(1024, 2, 32).astype(np.float64)
    live = a[:, 0, 0] > 0
    a = a[live]
    w = plan.meta[0]['desc']
    print('hw %d %d->%d B %d  %s ksplit %d  events %.1f us for %s; %d workgroups stamped' % (hw, cin, cout, B, plan.kernel_name(0), w.ksplit, e0.elapsed_time(e1) * 1e3, [o[0] for o in plan.ops], len(a)))
    t0 = a[:, 0, 0].min()
    us = lambda v: (v - t0) / 100.0
    q = lambda v: '%6.2f %6.2f %6.2f' % (np.min(v), np.median(v), np.max(v))
    print('   [us since the first workgroup started: min median max over workgroups]')
    print('   entry            ', q(us(a[:, 0, 0])))
    print('   set-up done      ', q(us(a[:, 0, 1])))
    print('   loader set-up    ', q(us(a[:, 1, 0])))
    nt = int(((a[0, 0, 4:] > 0).sum()) // 2)
    for i in range(min(2, int((a[0, 1, 1:] > 0).sum()))):
        print('   loader issue %d   ' % i, q(us(a[:, 1, 1 + i])))
    for t in range(min(nt, 6)):
        print('   tile %d landed    ' % t, q(us(a[:, 0, 4 + 2 * t])), '  computed', q(us(a[:, 0, 5 + 2 * t])), '  (compute %s)' % q((a[:, 0, 5 + 2 * t] - a[:, 0, 4 + 2 * t]) / 100.0))
    print('   walk done        ', q(us(a[:, 0, 2])), ' tiles per workgroup', nt)
    print('   flush done       ', q(us(a[:, 0, 3])), '  (flush %s)' % q((a[:, 0, 3] - a[:, 0, 2]) / 100.0))


for arg in sys.argv[1:] or ['10,512,512', '38,64,64', '125,64,64']:
    run(*[int(v) for v in arg.split(',')])
