#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "conv_first_gen or deconv or Deconv or abi or precision" > gpurun_out/ab30_tests.txt 2>&1 || { tail -40 gpurun_out/ab30_tests.txt; exit 1; }
tail -2 gpurun_out/ab30_tests.txt
timeout -k 10 300 python tools/op_table.py --model deconv --size 512 --classes 2 > gpurun_out/op_deconv.txt 2>/dev/null; tail -3 gpurun_out/op_deconv.txt
timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null
for f in 0 1; do SEG_FIRST_GEN=$f timeout -k 10 200 python bench.py --model deconv --size 512 --classes 2 --steps 20 --warmup 5 --no-cpu-baseline --mode infer 2>/dev/null; done
