set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_bwd or fused_maxpool" > gpurun_out/r04_tests2.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests2.log
tail -15 gpurun_out/r04_tests2.log
grep -q "tests rc 0" gpurun_out/r04_tests2.log || exit 1
timeout -k 10 300 python tools/conv_micro2.py --size 512 --cfgs 0,104,208,204 > gpurun_out/r04_ring_micro512.txt 2>&1; cat gpurun_out/r04_ring_micro512.txt
timeout -k 10 300 python tools/conv_micro2.py --size 256 --cfgs 0,104,208,204 > gpurun_out/r04_ring_micro256.txt 2>&1; cat gpurun_out/r04_ring_micro256.txt
timeout -k 10 300 python tools/stamp_ring.py 58,256,256,16,208 58,256,256,16,204 123,128,128,16,208 26,256,256,16,204 > gpurun_out/r04_stamp_ring.log 2>&1; cat gpurun_out/r04_stamp_ring.log
