"""Per-launch durations of the kernels whose name contains argv[2], from a rocprofv3 --kernel-trace CSV (argv[1])."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if sys.argv[2] in n:
        agg[n[:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in agg.items():
    print(n, 'launches', len(v), 'last 16 [us]:', ' '.join('%.0f' % d for d in v[-16:]))
