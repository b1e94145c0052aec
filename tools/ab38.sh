#!/bin/bash
# thin 3x3 kernels with all nine loads issued before the arithmetic (no per-tap branch)
mkdir -p gpurun_out; L=gpurun_out/ab38.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "on_load or thin or deconv or Deconv or advers" > gpurun_out/ab38_tests.txt 2>&1 || { tail -40 gpurun_out/ab38_tests.txt; exit 1; }
tail -2 gpurun_out/ab38_tests.txt
for r in 1 2; do
echo "deconv512 train" >> $L; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
echo "deconv512 infer" >> $L; timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done
cat $L
timeout -k 10 300 python tools/op_table.py --model deconv --size 512 --classes 2 2>/dev/null | grep -i "conv_out\|sum per"
