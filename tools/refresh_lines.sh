#!/bin/bash
# GPU box: only the bench lines of tools/the round profiles (profiles/r03_bench_lines.jsonl)
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/r03_bench_lines.jsonl; rm -f $L
echo line; timeout -k 10 200 python bench.py >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --host-data --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --size 512 --steps 20 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --nk 64 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode mc --batch 32 --steps 5 --warmup 2 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode infer --batch 32 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --mode infer --size 512 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 10 --warmup 3 --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --mode infer --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --adversarial --no-cpu-baseline >> $L 2>/dev/null
echo line; timeout -k 10 200 python bench.py --size 512 --steps 20 --warmup 5 --adversarial --no-cpu-baseline >> $L 2>/dev/null
wc -l $L
