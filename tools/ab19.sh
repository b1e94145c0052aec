#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run new X=0
run red15 SEG_SWEEP_REDUCE_US=15
run red30 SEG_SWEEP_REDUCE_US=30
run red60 SEG_SWEEP_REDUCE_US=60
run batched SEG_REDUCE_FLAVOR=batched
run batched_red30 SEG_REDUCE_FLAVOR=batched SEG_SWEEP_REDUCE_US=30
run new2 X=0
