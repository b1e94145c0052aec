#!/usr/bin/env python
"""Debug build only: the s_memrealtime stamps of ONE layer's wgrad_sweep_kernel launch INSIDE the eager C2 train step (next to the
data gradients and the other side stream's filter gradient), beside the same launch alone on the chip.
    python tools/stamp_step.py [layer=conv3_2] [size=256] [batch=16]"""
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
from segmentation_amd import _lib as L
from segmentation_amd.datasets import SyntheticDataSet
from segmentation_amd.unet import UNetModel
lib = L.load()
lib.seg_dbg_set_swstamps.argtypes = [C.c_void_p]; lib.seg_dbg_set_swstamps.restype = C.c_int
lib.seg_dbg_set_swfilter.argtypes = [C.c_int, C.c_int, C.c_int]; lib.seg_dbg_set_swfilter.restype = C.c_int
layer = sys.argv[1] if len(sys.argv) > 1 else 'conv3_2'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ds = SyntheticDataSet(B, size, 4)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)
m._bind_batch(ds)
for _ in range(5):
    m.train_step()
torch.cuda.synchronize()
plan = m.step_plan
idx = [i for i, (n, _, _) in enumerate(plan.ops) if n == layer + '/dw'][0]
w = plan.meta[idx]['desc']
kp = (w.src0.c + (w.src1.c if w.src1.ptr else 0))
assert lib.seg_dbg_set_swfilter(w.Ho, kp, w.dz.c) == 0
st = torch.zeros(1024 * 64, dtype=torch.int64, device='cuda')


def show(a, title):
    a = a.reshape(1024, 2, 32).astype(np.float64)
    a = a[a[:, 0, 0] > 0]
    t0 = a[:, 0, 0].min()
    us = lambda v: (v - t0) / 100.0
    q = lambda v: '%7.2f %7.2f %7.2f' % (np.min(v), np.median(v), np.max(v))
    nt = int(((a[0, 0, 4:] > 0).sum()) // 2)
    print('%s: %s ksplit %d, %d workgroups, %d tiles each   [us: min median max over workgroups]' % (title, plan.kernel_name(idx), w.ksplit, len(a), nt))
    print('   entry            ', q(us(a[:, 0, 0])))
    print('   set-up done      ', q(us(a[:, 0, 1])))
    print('   first tile landed', q(us(a[:, 0, 4])))
    per = (a[:, 0, 2] - a[:, 0, 4]) / 100.0 / max(nt, 1)
    print('   per tile (walk / tiles)', q(per))
    comp = np.stack([(a[:, 0, 5 + 2 * t] - a[:, 0, 4 + 2 * t]) / 100.0 for t in range(min(nt, 12))], 1)
    wait = np.stack([(a[:, 0, 4 + 2 * (t + 1)] - a[:, 0, 5 + 2 * t]) / 100.0 for t in range(min(nt, 12) - 1)], 1) if nt > 1 else np.zeros((len(a), 1))
    print('   compute per tile ', q(comp.mean(1)), '  wait at the barrier per tile', q(wait.mean(1)))
    print('   walk done        ', q(us(a[:, 0, 2])))
    print('   flush done       ', q(us(a[:, 0, 3])), ' (flush %s)' % q((a[:, 0, 3] - a[:, 0, 2]) / 100.0))


assert lib.seg_dbg_set_swstamps(st.data_ptr()) == 0
m.train_step(); torch.cuda.synchronize()
show(st.cpu().numpy(), 'in the step')
st.zero_()
name, fn, args = plan.ops[idx]
fn(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
show(st.cpu().numpy(), 'alone      ')
lib.seg_dbg_set_swstamps(None)
