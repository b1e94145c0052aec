#!/bin/bash
# A/B of step-structure switches on one box: each arm = one bench.py run (200 steps x 5 windows), prints value + windows
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run base SEG_EARLY_UPDATE=0
run early SEG_EARLY_UPDATE=1
run early_share SEG_EARLY_UPDATE=1 SEG_SHARE_AUX=1
run early_share_w3 SEG_EARLY_UPDATE=1 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3
run base_share_w3 SEG_EARLY_UPDATE=0 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3
run base_w3 SEG_EARLY_UPDATE=0 SEG_WGRAD_STREAMS=3
run base_w1 SEG_EARLY_UPDATE=0 SEG_WGRAD_STREAMS=1
