#!/bin/bash
# GPU box: refresh every committed round-2 profile / bench line with the current build
set -e
cd $GRAFT_REPO_ROOT
echo profiles r02; timeout -k 10 300 bash tools/collect_profiles.sh r02 > gpurun_out/refresh_r02.log 2>&1
echo r02_c4; timeout -k 10 300 bash tools/collect_profiles.sh r02_c4 --size 512 --steps 20 --warmup 5 >> gpurun_out/refresh_r02.log 2>&1
echo r02_c3; timeout -k 10 300 bash tools/collect_profiles.sh r02_c3 --model fcn8s --size 512 --classes 21 --batch 8 >> gpurun_out/refresh_r02.log 2>&1
echo r02_c5; timeout -k 10 300 bash tools/collect_profiles.sh r02_c5 --mode mc --batch 32 --steps 5 --warmup 2 >> gpurun_out/refresh_r02.log 2>&1
L=gpurun_out/r02_bench_lines.jsonl; rm -f $L
echo line; timeout -k 10 200 python bench.py >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --host-data --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --size 512 --steps 20 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --nk 64 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode mc --batch 32 --steps 5 --warmup 2 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode infer --batch 32 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode infer --size 512 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 10 --warmup 3 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --adversarial --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --size 512 --steps 20 --warmup 5 --adversarial --no-cpu-baseline >> $L 2>/dev/null
wc -l $L
