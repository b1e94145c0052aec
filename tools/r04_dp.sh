cd $GRAFT_REPO_ROOT
bash tools/run_gpu_tests.sh -k "dp or extras or unet" || exit 1
grep -q "tests rc 0" gpurun_out/gpu_tests.log || exit 1
( for a in "" "--force-dist" "--force-dist --force-collectives"; do for r in 1 2; do echo -n "args '$a': "; MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline --dp-cuts default 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['ms_per_step_windows']['median'])"; done; done ) > gpurun_out/r04_dp_overhead.txt 2>&1; cat gpurun_out/r04_dp_overhead.txt
