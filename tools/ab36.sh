#!/bin/bash
# conv1_0's filter gradient straight from the image (SEG_FIRST_GEN_WGRAD=1, default) against im2col + the 1x1 filter gradient
mkdir -p gpurun_out; L=gpurun_out/ab36.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "first_gen or deconv or Deconv or abi" > gpurun_out/ab36_tests.txt 2>&1 || { tail -40 gpurun_out/ab36_tests.txt; exit 1; }
tail -2 gpurun_out/ab36_tests.txt
for r in 1 2; do for f in 0 1; do
echo "first_gen_wgrad=$f deconv512 train" >> $L; SEG_FIRST_GEN_WGRAD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
timeout -k 10 300 python tools/op_table.py --model deconv --size 512 --classes 2 2>/dev/null | grep -i "conv1_0\|sum per"
