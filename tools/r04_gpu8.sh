cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_bwd or fused_maxpool or split_k or upconv" > gpurun_out/r04_tests2.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests2.log
tail -3 gpurun_out/r04_tests2.log
grep -q "tests rc 0" gpurun_out/r04_tests2.log || exit 1
bash tools/ab_flag.sh gldsb "-DSEG_GLDS_BUILTIN" 3 --steps 50 --warmup 10 --windows 3 2>&1 | tail -7; cp gpurun_out/ab_flag_gldsb.txt gpurun_out/r04_ab_glds_asm_256.txt
