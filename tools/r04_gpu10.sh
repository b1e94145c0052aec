cd $GRAFT_REPO_ROOT
ls -la segmentation_amd/build/libseg_base.so || exit 1
bash tools/ab_run.sh 3 --size 512 --steps 20 --warmup 5 --windows 3 --no-roofline 2>&1 | tail -6 > gpurun_out/r04_ab_wgrad32_512.txt; cat gpurun_out/r04_ab_wgrad32_512.txt
bash tools/ab_run.sh 3 --steps 50 --warmup 10 --windows 3 --no-roofline 2>&1 | tail -6 > gpurun_out/r04_ab_wgrad32_256.txt; cat gpurun_out/r04_ab_wgrad32_256.txt
