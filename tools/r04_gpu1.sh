set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r04_tests1.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests1.log
tail -5 gpurun_out/r04_tests1.log
python tools/calibrate.py > gpurun_out/r04_calibrate.log 2>&1; tail -30 gpurun_out/r04_calibrate.log
python tools/stamp_conv.py 58,256,256,16,104 58,256,256,16,106 123,128,128,16,104 48,512,256,16,104 > gpurun_out/r04_stamp_conv.log 2>&1; tail -40 gpurun_out/r04_stamp_conv.log
bash tools/r04_counters.sh > gpurun_out/r04_counters.log 2>&1; python tools/r04_counters.py > gpurun_out/r04_sq_counters.json; tail -5 gpurun_out/r04_counters.log
