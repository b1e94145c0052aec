#!/usr/bin/env python
"""Per-stream timeline of one train step from a rocprofv3 --kernel-trace CSV: for each HIP stream (queue) the busy time, the
idle gaps between consecutive kernels, and the step's critical intervals.
usage: timeline.py <dir with *_kernel_trace.csv> [steps_to_skip]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts at every step_begin_kernel
starts = [i for i, r in enumerate(rows) if 'step_begin' in r['Kernel_Name']]
if len(starts) < 12:
    print('too few steps', len(starts)); sys.exit(1)
sel = starts[len(starts) // 2: len(starts) // 2 + 8]
for a, b in zip(sel[:-1], sel[1:]):
    R = rows[a:b]
    t0 = int(R[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in R)
    per = collections.defaultdict(list)
    for r in R:
        per[r['Queue_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name']))
    print('step: %.1f us, %d kernels' % ((t1 - t0) / 1e3, len(R)))
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks) / 1e3
        gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
        print('  queue %s: %3d kernels, busy %7.1f us, span %7.1f..%7.1f us, gaps sum %6.1f us (max %5.1f, median %4.1f)' % (
            q, len(ks), busy, ks[0][0] / 1e3, ks[-1][1] / 1e3, sum(g for g in gaps if g > 0), max(gaps or [0]), sorted(gaps or [0])[len(gaps) // 2]))
if len(sys.argv) > 2:
    a, b = sel[0], sel[1]
    t0 = int(rows[a]['Start_Timestamp'])
    for r in rows[a:b]:
        print('%8.1f %8.1f  q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['Queue_Id'], r['Kernel_Name'][:70]))
