#!/bin/bash
# bn8 never materialised: conv_out and its filter gradient normalise deconv3_0's output on load (SEG_BN_ON_LOAD=1, default)
mkdir -p gpurun_out; L=gpurun_out/ab37.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "on_load or thin or deconv or Deconv or abi" > gpurun_out/ab37_tests.txt 2>&1 || { tail -40 gpurun_out/ab37_tests.txt; exit 1; }
tail -2 gpurun_out/ab37_tests.txt
for r in 1 2; do for f in 0 1; do
echo "bn_on_load=$f deconv512 train" >> $L; SEG_BN_ON_LOAD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
echo "bn_on_load=$f deconv512 infer" >> $L; SEG_BN_ON_LOAD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
