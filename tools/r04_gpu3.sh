set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_bwd or fused_maxpool" > gpurun_out/r04_tests2.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests2.log
tail -4 gpurun_out/r04_tests2.log
grep -q "tests rc 0" gpurun_out/r04_tests2.log || exit 1
timeout -k 10 300 python tools/conv_micro2.py --size 512 --cfgs 0,208,204 --layers ${LAYERS512:-conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_1,conv5_2,conv6_1,conv6_2,conv7_1,conv7_2,conv8_1,conv8_2} > gpurun_out/r04_ring_micro512.txt 2>&1; cat gpurun_out/r04_ring_micro512.txt
timeout -k 10 300 python tools/conv_micro2.py --size 256 --cfgs 0,208,204 --layers conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_2,conv6_1,conv7_1 > gpurun_out/r04_ring_micro256.txt 2>&1; cat gpurun_out/r04_ring_micro256.txt
timeout -k 10 300 python tools/stamp_ring.py 58,256,256,16,208 58,256,256,16,204 123,128,128,16,208 > gpurun_out/r04_stamp_ring.log 2>&1; cat gpurun_out/r04_stamp_ring.log
