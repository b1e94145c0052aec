#!/bin/bash
# alternates bench.py between the base library (tools/ab_build.sh) and the working-tree build on ONE box:
#   tools/ab_run.sh <rounds> <bench.py args...>
n=$1; shift
for i in $(seq $n); do
  for arm in base new; do
    if [ $arm = base ]; then export SEG_LIB_PATH=$PWD/segmentation_amd/build/libseg_base.so; else unset SEG_LIB_PATH; fi
    timeout -k 10 200 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_$arm.json 2>/dev/null || exit 1
    echo -n "$arm "; python tools/bench_line.py gpurun_out/ab_$arm.json | cut -c1-80
  done
done
