#!/bin/bash
# conv1_0 of the DeconvModel straight from the image (seg_conv_first_gen) against im2col + 1x1 convolution; where the filter gradient's im2col goes
mkdir -p gpurun_out; L=gpurun_out/ab29.txt; : > $L
for r in 1 2; do for f in 0 1 2 3; do
echo "gen=$f deconv512" >> $L; SEG_FIRST_GEN=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
