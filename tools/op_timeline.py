#!/usr/bin/env python
"""Timeline of ONE eager train step without a profiler: every launch of the single-GPU step plan bracketed by HIP events on the
stream it runs on (forks are events, as in Plan.run_profiled), start / end printed relative to the first launch.  The events cost a
few us per launch, so the step comes out 10-20 % longer than the timed one; what the table shows is where each stream waits.
    python tools/op_timeline.py [--size 256] [--batch 16] [--model unet] [--reps 5]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--classes', type=int, default=4)
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
from segmentation_amd import _lib as L            # noqa: E402
from segmentation_amd.datasets import SyntheticDataSet      # noqa: E402
from segmentation_amd.unet import UNetModel            # noqa: E402

ds = SyntheticDataSet(a.batch, a.size, a.classes)
m = UNetModel(sess=None, dataset=ds, n_classes=a.classes, input_dims=a.size, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', use_graph=False)
m._bind_batch(ds)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
plan, side, flavor = m.step_plan, m._side, m._flavor()
main = torch.cuda.current_stream()
sp = C.c_void_p(main.cuda_stream)
EV = lambda: torch.cuda.Event(enable_timing=True)


def one():
    recs, used, aux_used, aux = [], {}, False, side[-1]
    origin = EV(); origin.record(main)
    for i, (name, fn, args) in enumerate(plan.ops):
        md = plan.meta[i]
        tag = md.get('side', 0)
        if md.get('flavor', flavor) != flavor:
            continue
        if fn is None and name in ('join_all', 'join_wgrad'):
            for o_ in used.values():
                ev = torch.cuda.Event(); ev.record(o_); main.wait_event(ev)
            if aux_used and name == 'join_all':
                ev = torch.cuda.Event(); ev.record(aux); main.wait_event(ev)
            continue
        if fn is None:
            if md.get('marker'):
                continue
            if aux_used:
                ev = torch.cuda.Event(); ev.record(aux); main.wait_event(ev); aux_used = False
            continue
        if tag == 'aux':
            st = aux; aux_used = True
        elif tag:
            st = side[(tag - 1) % (len(side) - 1)] if len(side) > 1 else side[0]
            used[id(st)] = st
        else:
            st = main
        if st is not main:
            ev = torch.cuda.Event(); ev.record(main); st.wait_event(ev)
        e0, e1 = EV(), EV()
        e0.record(st)
        d_ = md.get('desc')
        if isinstance(d_, L.ConvDesc):
            d_.signal = None
        rc = fn(*args, C.c_void_p(st.cuda_stream) if st is not main else sp)
        assert rc == 0, name
        e1.record(st)
        recs.append((name, tag, e0, e1))
    for st in used.values():
        ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)
    torch.cuda.synchronize()
    return [(n, t, origin.elapsed_time(e0) * 1e3, origin.elapsed_time(e1) * 1e3) for n, t, e0, e1 in recs]


rows = None
for _ in range(a.reps):
    m.loss_buf.zero_()
    cur = one()
    rows = cur if rows is None else [(n, t, s0 + s1, e0 + e1) for (n, t, s0, e0), (_, _, s1, e1) in zip(rows, cur)]
col = {0: 0, 1: 1, 2: 2, 'aux': 3}
print('%-26s %-4s %9s %9s %8s' % ('op', 'strm', 'start', 'end', 'us'))
last = {}
for n, t, s, e in rows:
    s /= a.reps; e /= a.reps
    gap = s - last.get(t, 0.0)
    last[t] = e
    print('%-26s %-4s %9.1f %9.1f %8.1f   %s%s' % (n, t, s, e, e - s, '          ' * col.get(t, 0), '#' * max(1, int((e - s) / 4))) + ('   (idle %.0f)' % gap if gap > 8 else ''))
print('step end %.1f us' % max(e / a.reps for _, _, _, e in rows))
