#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run plain X=0
BENCH_ARGS="--force-dist" run dp X=0
BENCH_ARGS="--force-dist" run dp_wgs128 SEG_WGRAD_WGS=128
BENCH_ARGS="--force-dist" run dp_dry SEG_DP_DRY=1
BENCH_ARGS="--force-dist" run dp_cuts2 SEG_DP_CUTS=conv3_1
BENCH_ARGS="--force-dist" run dp_hiprio SEG_DP_HIPRIO=1
run plain2 X=0
