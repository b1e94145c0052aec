#!/bin/bash
# GPU box: the round-3 profile set (kernel stats + counter passes of C2 / C4 shard / C3, op tables); tools/refresh_lines.sh adds the bench lines
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1 || { echo "r03 failed"; tail -5 gpurun_out/collect_r03.log; exit 1; }
bash tools/collect_profiles.sh r03_c4 --size 512 --steps 20 --warmup 5 > gpurun_out/collect_r03_c4.log 2>&1 || { echo "c4 failed"; exit 1; }
bash tools/collect_profiles.sh r03_c3 --model fcn8s --size 512 --classes 21 --batch 8 > gpurun_out/collect_r03_c3.log 2>&1 || { echo "c3 failed"; exit 1; }
python tools/op_table.py --size 256 > profiles/r03_op_table_256.txt 2>/dev/null
python tools/op_table.py --size 512 > profiles/r03_op_table_512.txt 2>/dev/null
python tools/op_table.py --model fcn8s --size 512 --classes 21 --batch 8 > profiles/r03_op_table_fcn8s.txt 2>/dev/null
mkdir -p gpurun_out/profiles_out; cp profiles/r03_* profiles/pmc_summary.json gpurun_out/profiles_out/
ls gpurun_out/profiles_out
