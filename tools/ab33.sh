#!/bin/bash
# XCD-aware workgroup order of the thin 3x3 kernels on / off, stand-alone launches of the DeconvModel's conv_out in ONE box
mkdir -p gpurun_out; L=gpurun_out/ab33.txt; : > $L
for r in 1 2; do for f in 0 1; do
echo "remap=$f" >> $L; SEG_XCD_REMAP=$f timeout -k 10 300 python tools/op_table.py --model deconv --size 512 --classes 2 2>/dev/null | grep -i "conv_out" >> $L
done; done
for r in 1 2; do for f in 0 1; do
echo "remap=$f deconv512 train" >> $L; SEG_XCD_REMAP=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
