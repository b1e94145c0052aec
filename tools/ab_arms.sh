#!/bin/bash
# A/B of several environment settings on ONE box, arms interleaved round by round:
#   tools/ab_arms.sh <rounds> <out name> "<arm1: VAR=a VAR2=b>" "<arm2>" ... -- [bench.py args]
n=$1; out=$2; shift 2
arms=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do arms+=("$1"); shift; done; shift
mkdir -p gpurun_out; L=gpurun_out/ab_$out.txt; : > $L
for r in $(seq $n); do for a in "${arms[@]}"; do
  echo -n "[$a] " >> $L
  env $a timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['ms_per_step_windows']['median'])" >> $L || exit 1
done; done
cat $L
