#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
export STEPS=40 BENCH_ARGS="--model deconv --size 512 --classes 2"
run deconv X=0
run deconv_wgs256 SEG_WGRAD_WGS=256
run deconv_wgs64 SEG_WGRAD_WGS=64
run deconv_noshare SEG_SHARE_AUX=0
run deconv_nobalance SEG_WGRAD_BALANCE=0
