#!/bin/bash
# GPU box: linearised-tile tests, then A/B of the selection threshold (SEG_CONV_LIN_PCT: 0 = off) in the C2 and 512^2 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 400 -k "conv_fwd_bwd or tile_choice or splitk or fused_maxpool" > gpurun_out/lin_tests.log 2>&1 || { tail -30 gpurun_out/lin_tests.log; exit 1; }
tail -3 gpurun_out/lin_tests.log
bash tools/ab_env.sh 3 SEG_CONV_LIN_PCT "0 13 8 4" --size 256 && cp gpurun_out/ab_env_SEG_CONV_LIN_PCT.txt gpurun_out/ab_lin_256.txt
bash tools/ab_env.sh 2 SEG_CONV_LIN_PCT "0 13 4" --size 512 && cp gpurun_out/ab_env_SEG_CONV_LIN_PCT.txt gpurun_out/ab_lin_512.txt
