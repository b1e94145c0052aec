#!/usr/bin/env python
"""Analyse a rocprofv3 --kernel-trace CSV: per-step wall time, busy time (union of kernel intervals), per-kernel-family
sums and the list of kernels of one steady-state step in start order."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    for k in ('conv_fwd_glds_kernel', 'conv_fwd_kernel', 'conv_wgrad_kernel', 'wgrad_reduce_batch', 'wgrad_reduce_kernel', 'conv_first_mfma', 'conv_first_fwd', 'im2col', 'head_xent', 'step_begin', 'conv_ws_kernel', 'maxpool_fwd', 'maxpool_bwd',
              'softmax_xent', 'adam_kernel', 'pack_kernel', 'step_inc', 'bias_grad', 'copyBuffer', 'FillFunctor'):
        if k in n: return k
    return n[:40]
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = adam[-3], adam[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
fam = collections.Counter(); cnt = collections.Counter()
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); fam[short(r['Kernel_Name'])] += d; cnt[short(r['Kernel_Name'])] += 1
print('step wall %.1f us, GPU busy (union) %.1f us, sum of kernel durations %.1f us, kernels %d' % ((t1 - t0) / 1e3, busy / 1e3, sum(fam.values()) / 1e3, len(step)))
for k, v in fam.most_common(): print('  %-24s %4d launches %8.1f us' % (k, cnt[k], v / 1e3))
if len(sys.argv) > 2:
    for r in step:
        print('%9.1f %8.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, short(r['Kernel_Name'])))
