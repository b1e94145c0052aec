#!/bin/bash
# batch norm + max-pool: the fused backward (SEG_BN_POOL_BWD=1, default) against pool backward + batch-norm backward (SEG_BN_POOL=0/1: the forward)
mkdir -p gpurun_out; L=gpurun_out/ab34.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "one_pass or deconv or Deconv or abi" > gpurun_out/ab34_tests.txt 2>&1 || { tail -40 gpurun_out/ab34_tests.txt; exit 1; }
tail -2 gpurun_out/ab34_tests.txt
for r in 1 2; do for f in 0 1; do
echo "bn_pool_bwd=$f deconv512 train" >> $L; SEG_BN_POOL_BWD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
echo "bn_pool_bwd=$f deconv512 infer" >> $L; SEG_BN_POOL_BWD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
