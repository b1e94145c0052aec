#!/bin/bash
# GPU box: what the data-parallel step costs on ONE GPU (bench.py --force-dist, eager launches): plain step / data-parallel walk with the
# one-rank collectives skipped (what a world-1 run does) / with RCCL's one-rank kernels forced; SEG_PLAN_C=0 = the Python walk
cd $GRAFT_REPO_ROOT
( for pc in 1 0; do for a in "" "--force-dist" "--force-dist --force-collectives"; do for r in 1 2; do echo -n "SEG_PLAN_C=$pc args '$a': "; SEG_PLAN_C=$pc MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout -k 10 300 python bench.py $a --no-graph --no-cpu-baseline --no-roofline --dp-cuts default 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['ms_per_step_windows']['median'], (d['config'].get('allreduce') or {}).get('exposed_us'))"; done; done; done ) > gpurun_out/r04_dp_overhead.txt 2>&1; cat gpurun_out/r04_dp_overhead.txt
