#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run base X=0
run side_low SEG_SIDE_PRIO=1
run side_low_wgs128 SEG_SIDE_PRIO=1 SEG_WGRAD_WGS=128
run side_low_wgs96 SEG_SIDE_PRIO=1 SEG_WGRAD_WGS=96
run side_high SEG_SIDE_PRIO=-1
run base2 X=0
