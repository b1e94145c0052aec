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
x = torch.zeros(1 << 20, device=dev)          # 4 MB: x.add_(1) is a ~5 us kernel
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
