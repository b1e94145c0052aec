#!/bin/bash
# conv_out's filter gradient on the vector ALU (SEG_THIN_WGRAD=1, default) against the MFMA walk over 32 x 32 padded channels
mkdir -p gpurun_out; L=gpurun_out/ab32.txt; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "thin or abi" > gpurun_out/ab32_tests.txt 2>&1 || { tail -40 gpurun_out/ab32_tests.txt; exit 1; }
tail -2 gpurun_out/ab32_tests.txt
for r in 1 2; do for f in 0 1; do
echo "thin_wgrad=$f deconv512 train" >> $L; SEG_THIN_WGRAD=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $L
done; done
cat $L
timeout -k 10 300 python tools/op_table.py --model deconv --size 512 --classes 2 2>/dev/null | grep -i "conv_out\|sum per"
