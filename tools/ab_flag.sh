#!/bin/bash
# A/B of a compile-time flag on ONE box: builds segmentation_amd/build/libseg_<tag>.so from the working tree with extra hipcc flags
# and alternates bench.py between it (arm "flag") and the default build (arm "default").
#   tools/ab_flag.sh <tag> "<flags>" <rounds> [bench.py args...]
set -e
cd "$(dirname "$0")/.."
tag=$1; flags=$2; n=$3; shift 3
d=segmentation_amd/build/ab_$tag; mkdir -p $d
python -c "from segmentation_amd import _build; _build.build(verbose=False)"
objs=""
for src in segmentation_amd/csrc/*.hip; do
  f=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -w $flags -Iinclude -c $src -o $d/$f.o &
  objs="$objs $d/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o segmentation_amd/build/libseg_$tag.so $objs
mkdir -p gpurun_out; L=gpurun_out/ab_flag_$tag.txt; : > $L
for i in $(seq $n); do
  for arm in flag default; do
    if [ $arm = flag ]; then export SEG_LIB_PATH=$PWD/segmentation_amd/build/libseg_$tag.so; else unset SEG_LIB_PATH; fi
    echo -n "$arm " >> $L
    timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['ms_per_step_windows']['median'])" >> $L || exit 1
  done
done
cat $L
