#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run base X=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run tuned SEG_CFG_OVERRIDE="conv8_1=12,conv7_1/dx01=1"
run base2 X=0
