#!/usr/bin/env python
"""What an event record / cross-stream wait costs on the GPU timeline: N back-to-back launches of a ~6 us kernel on a stream,
(a) bare, (b) with an event recorded on the stream after every launch, (c) with that event also waited for by a second stream
that launches a kernel, (d) with the stream itself waiting for an event of the second stream before every launch."""
import sys, time, torch
sys.path.insert(0, '.')
from segmentation_amd import _lib as L
import ctypes as C
lib = L.load()
dev = torch.device('cuda', 0)
N = 400
SZ = int(sys.argv[1]) if len(sys.argv) > 1 else 24        # log2 elements: 2^24 floats = 64 MB, x.add_(1) ~ 25-30 us (GPU-bound loop); 20 = host-bound
x = torch.zeros(1 << SZ, device=dev)
y = torch.zeros(1 << 20, device=dev)
def run(mode, prio):
    a = torch.cuda.Stream(dev, priority=prio); b = torch.cuda.Stream(dev)
    def body():
        for i in range(N):
            with torch.cuda.stream(a):
                if mode == 'wait_other':
                    ev2 = torch.cuda.Event(); ev2.record(b); a.wait_event(ev2)
                x.add_(1)
                if mode in ('record', 'fork'):
                    ev = torch.cuda.Event(); ev.record(a)
            if mode == 'fork':
                b.wait_event(ev)
                with torch.cuda.stream(b):
                    y.add_(1)
    body(); torch.cuda.synchronize()
    t0 = time.perf_counter(); body(); t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print('%-11s priority %2d : %6.2f us per iteration on the GPU timeline (host issue %5.2f us)' % (mode, prio, t / N * 1e6, t_issue / N * 1e6))
for prio in (0, -1):
    for mode in ('bare', 'record', 'fork', 'wait_other'):
        run(mode, prio)

# stream memory operations as a cheaper fork?  main: hipStreamWriteValue32(flag, i) after each kernel; side: hipStreamWaitValue32(flag >= i)
h = L.hip_runtime()
def run_value(prio, with_wait):
    a = torch.cuda.Stream(dev, priority=prio); b = torch.cuda.Stream(dev)
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    h.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
    h.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
    base = [0]
    def body():
        for i in range(N):
            base[0] += 1
            with torch.cuda.stream(a):
                x.add_(1)
            rc = h.hipStreamWriteValue32(C.c_void_p(a.cuda_stream), C.c_void_p(flag.data_ptr()), base[0], 0)
            assert rc == 0, rc
            if with_wait:
                rc = h.hipStreamWaitValue32(C.c_void_p(b.cuda_stream), C.c_void_p(flag.data_ptr()), base[0], 0, 0xffffffff)   # 0 = hipStreamWaitValueGte
                assert rc == 0, rc
                with torch.cuda.stream(b):
                    y.add_(1)
    body(); torch.cuda.synchronize()
    t0 = time.perf_counter(); body(); t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print('%-11s priority %2d : %6.2f us per iteration on the GPU timeline (host issue %5.2f us)' % ('writevalue' + ('+wait' if with_wait else ''), prio, t / N * 1e6, t_issue / N * 1e6))
try:
    for prio in (0, -1):
        run_value(prio, False); run_value(prio, True)
except Exception as e:
    print('stream memory operations unavailable:', repr(e)[:200])
