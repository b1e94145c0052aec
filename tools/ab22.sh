#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run splitk X=0
run nosplit SEG_CONV_SPLITK=0
run splitk2 SEG_CONV_SPLITK=2
run splitk8 SEG_CONV_SPLITK=8
run splitk_b X=0
run nosplit_b SEG_CONV_SPLITK=0
