#!/bin/bash
# GPU box: per-kernel totals of one bench.py workload (rocprofv3 --kernel-trace --stats): tools/kstats2.sh <tag> <bench args...>
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/err
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-84s calls %6s avg %8.2f us  %5.1f %%' % (n[:84], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
tail -1 $O/bench.json | cut -c1-160
