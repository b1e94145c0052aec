#!/bin/bash
# A/B: softmax(logits) for the adversary written by the x-entropy launch (SEG_ADV_FUSE_PROBS=1, default) or by its own launch
mkdir -p gpurun_out; L=gpurun_out/ab28.txt; : > $L
python -m pytest tests -x -q -m gpu -k "advers" > gpurun_out/ab28_tests.txt 2>&1 || { tail -30 gpurun_out/ab28_tests.txt; exit 1; }
tail -2 gpurun_out/ab28_tests.txt
for r in 1 2; do for f in 0 1; do
echo "fuse=$f fcn8s" >> $L; SEG_ADV_FUSE_PROBS=$f timeout -k 10 200 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --adversarial --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
echo "fuse=$f unet512" >> $L; SEG_ADV_FUSE_PROBS=$f timeout -k 10 200 python bench.py --size 512 --steps 20 --warmup 5 --adversarial --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
