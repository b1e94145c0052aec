#!/usr/bin/env python
"""Debug build only: where a conv_ring_kernel launch spends its time (s_memrealtime stamps of thread 0 of every workgroup, 10 ns ticks:
entry, then per chunk "barrier passed", per tile "last chunk computed" and "epilogue done").
    python tools/stamp_ring.py hw,cin,cout[,B[,cfg]] ..."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from segmentation_amd import _build
XF = os.environ.get('STAMP_FLAGS', '').split()           # e.g. -DSEG_RING_NOREAD / -DSEG_RING_NOMMA (ablation builds: results are garbage)
LIBS = os.path.join(ROOT, 'segmentation_amd', 'build', 'libseg_rstamps%s.so' % ''.join(f.replace('-D', '_') for f in XF))
if 'SEG_LIB_PATH' not in os.environ:
    _build.build(verbose=False)
    d = os.path.join(ROOT, 'segmentation_amd', 'build')
    o = os.path.join(d, 'conv_ring_stamps.o')
    subprocess.check_call([_build.HIPCC] + _build.FLAGS + ['-DSEG_STAMPS'] + XF + ['-c', os.path.join(_build.CSRC, 'conv_ring.hip'), '-o', o])
    objs = [os.path.join(d, f.replace('.hip', '.o')) for f in _build.SOURCES if f != 'conv_ring.hip'] + [o]
    subprocess.check_call([_build.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIBS] + objs)
    os.environ['SEG_LIB_PATH'] = LIBS
    sys.exit(subprocess.call([sys.executable] + sys.argv))         # a child process with SEG_LIB_PATH set (never exec)
import ctypes as C, numpy as np, torch
from segmentation_amd import _lib as L, engine as E
lib = L.load()
lib.seg_dbg_set_crstamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_crstamps.restype = C.c_int


def run(hw, cin, cout, B=16, cfg=208):
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
    st = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
    assert lib.seg_dbg_set_crstamps(st.data_ptr()) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(s); e1.record(); torch.cuda.synchronize()
    lib.seg_dbg_set_crstamps(None)
    raw = st.cpu().numpy().reshape(512, 64).astype(np.float64)
    raw = raw[raw[:, 0] > 0]
    a = raw[:, :32]
    # in-kernel shader clock: s_memtime ticks per s_memrealtime tick (100 MHz), between the first and the last stamp of a workgroup
    clk = []
    for w in raw:
        n = int((w[:32] > 0).sum())
        if n >= 3:
            clk.append((w[32 + n - 1] - w[32 + 1]) / max(1.0, (w[n - 1] - w[1])) * 100.0)
    print('   in-kernel clock (median over workgroups): %.0f MHz' % np.median(clk))
    nch = (cin + 31) // 32
    print('hw %d %d->%d B %d  %s  events %.1f us; %d workgroups stamped, %d chunks per tile' % (hw, cin, cout, B, plan.kernel_name(0), e0.elapsed_time(e1) * 1e3, len(a), nch))
    t0 = a[:, 0].min()
    for wsel in (0, len(a) // 2, len(a) - 1):
        w = a[wsel]
        us = [(v - t0) / 100.0 for v in w if v > 0]
        line = '   wg %3d: entry %.2f |' % (wsel, us[0]); i = 1; k = 0
        while i < len(us) and k < 4:
            ch = us[i:i + nch]; i += nch
            line += ' tile%d chunks@ %s' % (k, ' '.join('%.2f' % v for v in ch))
            if i < len(us): line += ' computed %.2f' % us[i]; i += 1
            if i < len(us): line += ' stored %.2f |' % us[i]; i += 1
            k += 1
        print(line)
    ends = np.array([max(v for v in w if v > 0) for w in a]); print('   last stamp per workgroup: min %.2f median %.2f max %.2f us' % ((ends.min() - t0) / 100, (np.median(ends) - t0) / 100, (ends.max() - t0) / 100))


for arg in sys.argv[1:] or ['58,256,256,16,208', '58,256,256,16,204']:
    run(*[int(v) for v in arg.split(',')])
