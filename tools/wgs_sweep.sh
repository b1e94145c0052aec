#!/bin/bash
# step time against the workgroup target of the filter-gradient launches (SEG_WGRAD_WGS); usage: tools/wgs_sweep.sh [bench.py args]
cd "$(dirname "$0")/.."
for w in 32 48 64 96 128 192 256; do
  echo -n "SEG_WGRAD_WGS=$w: "; SEG_WGRAD_WGS=$w timeout -k 10 200 python bench.py --no-graph --steps 50 --warmup 20 --windows 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['config']['ms_per_step_windows']['all'])"
done
