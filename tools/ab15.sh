#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run base X=0
for w in 2 3; do for g in 48 64 80 96 128; do run share_w${w}_wgs$g SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=$w SEG_WGRAD_WGS=$g; done; done
run base2 X=0
