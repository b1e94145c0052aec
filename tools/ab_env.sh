#!/bin/bash
# One parameterised A/B of an environment switch on ONE box (replaces the r03 one-off ab13..ab40 scripts):
#   tools/ab_env.sh <rounds> <VAR> "<value1> <value2> ..." [bench.py args...]
# e.g.  tools/ab_env.sh 3 SEG_WGRAD_WGS "48 64 96" --size 256
# Arms are interleaved round by round (boxes differ by +-2 %, so arms are only comparable inside one call); '-' = variable unset.
n=$1; var=$2; vals=$3; shift 3
mkdir -p gpurun_out; L=gpurun_out/ab_env_${var}.txt; : > $L
for r in $(seq $n); do for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  echo -n "$var=$v " >> $L
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['ms_per_step_windows']['median'])" >> $L || exit 1
done; done
cat $L
