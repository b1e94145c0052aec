#!/usr/bin/env python
"""Shader clock DURING the train step: a one-lane sampler kernel (tools/micro/clockprobe.hip) runs on a stream of its own beside
back-to-back train steps and records (s_memtime, s_memrealtime) every 10 us.  usage: clock_probe.py [size] [batch]
Prints the clock's median / min / max over the steps and, for comparison, beside an idle GPU and beside a bare bf16 matmul loop."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
so = '/tmp/libclockprobe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(ROOT, 'tools/micro/clockprobe.hip'), '-o', so])
lib = C.CDLL(so)
lib.clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
from segmentation_amd.unet import UNetModel
from segmentation_amd.datasets import SyntheticDataSet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
probe_stream = torch.cuda.Stream(dev)


def sample(work, ms, label):
    n = int(ms * 100)                       # one sample per 10 us
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert lib.clock_probe(buf.data_ptr(), n, 1000, probe_stream.cuda_stream) == 0
    work()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(n, 2).astype(np.float64)
    dt, dr = np.diff(a[:, 0]), np.diff(a[:, 1])
    mhz = dt / dr * 100.0
    mhz = mhz[5:]                            # (the first samples are taken before the work has started)
    print('%-34s clock MHz: median %.0f  p10 %.0f  p90 %.0f  min %.0f  max %.0f   (%d samples over %.2f ms)' % (
        label, np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90), mhz.min(), mhz.max(), len(mhz), (a[-1, 1] - a[0, 1]) / 1e5))
    return mhz


sample(lambda: None, 1.0, 'idle GPU')
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); y = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    x @ y
sample(lambda: [x @ y for _ in range(12)], 8.0, 'hipBLASLt 8192^3 bf16 matmuls')
m = UNetModel(sess=None, dataset=SyntheticDataSet(B, S, 4, seed=5555, n_batches=2), n_classes=4, input_dims=S, learning_rate=1e-4, log_dir=None,
              save_dir=None, load_snapshot=False, dtype='bf16', n_kernels=32, seed=5555, use_graph=False)      # (bench.py's model and data)
for _ in range(30):
    m.train_step()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(50):
    m.train_step()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 50
print('U-Net %d^2 x %d train step: %.3f ms (no sampler)' % (S, B, ms))
nsteps = 60 if S <= 256 else 20
mhz = sample(lambda: [m.train_step() for _ in range(nsteps)], ms * nsteps * 1.05, 'U-Net %d^2 x %d train steps' % (S, B))
per = int(ms * 100)
# clock along one step: samples folded onto the step period (approximate: the sampler is not phase-locked)
k = (len(mhz) // per) * per
if k >= per * 4:
    f = mhz[:k].reshape(-1, per)
    prof = np.median(f, axis=0)
    print('median clock by tenth of the step:', ' '.join('%.0f' % v for v in [np.median(prof[i * per // 10:(i + 1) * per // 10]) for i in range(10)]))
