#!/bin/bash
# batch-norm statistics from the producing launch (SEG_BN_FUSE_STATS=1, default) against the separate statistics pass
mkdir -p gpurun_out; L=gpurun_out/ab31.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "statistics_from or conv_first_gen or deconv or Deconv or abi or thin" > gpurun_out/ab31_tests.txt 2>&1 || { tail -40 gpurun_out/ab31_tests.txt; exit 1; }
tail -2 gpurun_out/ab31_tests.txt
for r in 1 2; do for f in 0 1; do
echo "fuse=$f deconv512 train" >> $L; SEG_BN_FUSE_STATS=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
echo "fuse=$f deconv512 infer" >> $L; SEG_BN_FUSE_STATS=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
