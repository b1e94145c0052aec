#!/bin/bash
# On the GPU box: per-kernel average durations of a short default bench run (rocprofv3 --kernel-trace --stats), for the kernels matching $1.
pat=${1:-.}
O=$GRAFT_REPO_ROOT/gpurun_out/kstats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/err
python3 - "$O" "$pat" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if re.search(sys.argv[2], n):
        print('%-70s calls %5s avg %8.2f us' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -1 $O/bench.json | cut -c1-120
