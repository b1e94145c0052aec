#!/bin/bash
# pixels per lane group of the fused head (seg_head_xent) at C2: 4 (default) / 2 / 1
mkdir -p gpurun_out; L=gpurun_out/ab40.txt; : > $L
for r in 1 2 3; do for f in 4 2 1; do
echo "head_px=$f" >> $L; SEG_HEAD_PX=$f timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['ms_per_step_windows']['median'])" >> $L
done; done
cat $L
