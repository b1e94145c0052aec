cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r04_tests3.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests3.log
tail -6 gpurun_out/r04_tests3.log
grep -q "tests rc 0" gpurun_out/r04_tests3.log || exit 1
( for v in 1 0; do echo "SEG_PLAN_C=$v"; SEG_PLAN_C=$v timeout -k 10 200 python tools/cpu_issue.py; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_cpu_issue.txt; cat gpurun_out/r04_cpu_issue.txt
bash tools/ab_env.sh 3 SEG_SPLIT_PACK "0 1" --steps 50 --warmup 10 --windows 3 2>&1 | tail -7; cp gpurun_out/ab_env_SEG_SPLIT_PACK.txt gpurun_out/r04_ab_split_pack_256.txt
bash tools/ab_env.sh 2 SEG_PLAN_C "0 1" --steps 50 --warmup 10 --windows 3 2>&1 | tail -5; cp gpurun_out/ab_env_SEG_PLAN_C.txt gpurun_out/r04_ab_plan_c_256.txt
bash tools/ab_env.sh 2 SEG_SPLIT_PACK "0 1" --size 512 --steps 20 --warmup 5 --windows 3 2>&1 | tail -5; cp gpurun_out/ab_env_SEG_SPLIT_PACK.txt gpurun_out/r04_ab_split_pack_512.txt
