#!/bin/bash
# one box, alternating: bench.py under a list of environment settings ("NAME=VALUE" or "-" for the default), N rounds
#   tools/env_sweep.sh <rounds> "<setting> <setting> ..." <bench.py args...>
n=$1; shift; settings=$1; shift
for i in $(seq $n); do
  for s in $settings; do
    if [ "$s" = "-" ]; then pre=""; else pre="$s"; fi
    env $pre timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-roofline > gpurun_out/sweep.json 2>/dev/null || { echo "$s failed"; continue; }
    echo -n "$s  "; python tools/bench_line.py gpurun_out/sweep.json | cut -c41-62
  done
done
