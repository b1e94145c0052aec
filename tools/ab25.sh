#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run plain X=0
export BENCH_ARGS="--force-dist"
run dp X=0
run dp_share SEG_SHARE_AUX=1
run dp_share_wgs128 SEG_SHARE_AUX=1 SEG_WGRAD_WGS=128
run dp_share_wgs96 SEG_SHARE_AUX=1 SEG_WGRAD_WGS=96
run dp_wgs96 SEG_WGRAD_WGS=96
run dp_wgs192 SEG_WGRAD_WGS=192
run dp_main_join SEG_DP_JOIN=main
