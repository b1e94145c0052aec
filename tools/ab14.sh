#!/bin/bash
# A/B of stream-layout switches on one box: each arm = one bench.py run, prints value + windows.  BENCH_ARGS adds bench.py flags.
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; env "$@" python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/ab_$label.json; python -c "
import json,sys
j=json.load(open('gpurun_out/ab_$label.json')); print('%-28s' % '$label', j['value'], j['config']['ms_per_step_windows']['all'])"; }
run base X=0
run share_w3 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3
run share_w3_wgs64 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=64
run share_w3_wgs96 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=96
run share_w3_wgs160 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=160
run share_w3_wgs256 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=256
run share_w4 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=4
run share_w2 SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=2
run share_w3_nohiprio SEG_SHARE_AUX=1 SEG_WGRAD_STREAMS=3 SEG_HIPRIO=0
