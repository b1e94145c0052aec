#!/bin/bash
# bn_final with 128 row slices per workgroup (was 32)
mkdir -p gpurun_out; L=gpurun_out/ab39.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "batch_norm or one_pass or on_load or deconv or Deconv or statistics" > gpurun_out/ab39_tests.txt 2>&1 || { tail -40 gpurun_out/ab39_tests.txt; exit 1; }
tail -2 gpurun_out/ab39_tests.txt
for r in 1 2; do
echo "deconv512 train" >> $L; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['families'].get('bn_final_kernel'))" >> $L
echo "deconv512 infer" >> $L; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done
cat $L
