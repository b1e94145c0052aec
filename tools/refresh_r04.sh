#!/bin/bash
# GPU box: the round-4 profile set -- kernel stats + counter passes of C2 / C4 shard, op tables, bench lines (bf16 and f32), the
# per-queue timeline of a replayed hipGraph step against the eager step.  Outputs under gpurun_out/ and profiles/.
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1 || { echo "r04 failed"; tail -5 gpurun_out/collect_r04.log; exit 1; }
cp profiles/r04_pmc_summary.json profiles/pmc_summary.json
bash tools/collect_profiles.sh r04_c4 --size 512 --steps 20 --warmup 5 > gpurun_out/collect_r04_c4.log 2>&1 || { echo "c4 failed"; exit 1; }
python tools/op_table.py --size 256 2>/dev/null | grep -v "Training from" > profiles/r04_op_table_256.txt
python tools/op_table.py --size 512 2>/dev/null | grep -v "Training from" > profiles/r04_op_table_512.txt
L=gpurun_out/r04_bench_lines.jsonl; rm -f $L
timeout -k 10 300 python bench.py >> $L 2>/dev/null
timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --dtype f32 --size 512 --steps 4 --warmup 2 --windows 2 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --host-data --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --mode mc --batch 32 --steps 5 --warmup 2 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --mode infer --batch 32 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --mode infer --size 512 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --model deconv --size 512 --classes 2 --steps 10 --warmup 3 --no-cpu-baseline >> $L 2>/dev/null
timeout -k 10 300 python bench.py --model fcn8s --size 512 --classes 21 --batch 8 --adversarial --no-cpu-baseline >> $L 2>/dev/null
wc -l $L; cp $L profiles/r04_bench_lines.jsonl
# world-1 data-parallel step: plain / DP path without collectives / DP path with RCCL's one-rank kernels
( for a in "" "--force-dist" "--force-dist --force-collectives"; do for r in 1 2; do echo -n "args '$a': "; MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline --dp-cuts default 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['ms_per_step_windows']['median'])"; done; done ) > profiles/r04_dp_overhead.txt 2>&1; cat profiles/r04_dp_overhead.txt
# graph replay against eager launches: per-queue timelines of a C2 step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl_graph gpurun_out/tl_eager
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_graph -- python3 bench.py --graph --steps 30 --warmup 10 --windows 1 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/tl_graph.err
SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_eager -- python3 bench.py --no-graph --steps 30 --warmup 10 --windows 1 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/tl_eager.err
( echo "== hipGraph replay (bench.py --graph)"; python tools/timeline.py gpurun_out/tl_graph | head -24; echo "== eager launches through seg_plan_run (bench.py --no-graph)"; python tools/timeline.py gpurun_out/tl_eager | head -24 ) > profiles/r04_graph_vs_eager_timeline.txt 2>&1
head -30 profiles/r04_graph_vs_eager_timeline.txt
mkdir -p gpurun_out/profiles_out; cp profiles/r04_* profiles/pmc_summary.json gpurun_out/profiles_out/
