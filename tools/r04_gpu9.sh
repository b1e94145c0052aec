cd $GRAFT_REPO_ROOT
( echo "== size 512 (default target 128)"; timeout -k 10 300 python tools/wgrad_micro.py --size 512 --cfgs 0,3,9 --ks 0 --layers conv9_1,conv9_2,conv1_2,conv2_1
  echo "== size 512 SEG_WGRAD_IMPL=old auto"; SEG_WGRAD_IMPL=old timeout -k 10 300 python tools/wgrad_micro.py --size 512 --cfgs 0 --ks 0 --layers conv9_1,conv9_2,conv1_2,conv2_1
  echo "== size 256 target 64"; SEG_WGRAD_WGS=64 timeout -k 10 300 python tools/wgrad_micro.py --size 256 --cfgs 0,3,9 --ks 0 --layers conv9_1,conv9_2,conv1_2,conv2_1
  echo "== size 256 target 64 SEG_WGRAD_IMPL=old auto"; SEG_WGRAD_IMPL=old SEG_WGRAD_WGS=64 timeout -k 10 300 python tools/wgrad_micro.py --size 256 --cfgs 0 --ks 0 --layers conv9_1,conv9_2,conv1_2,conv2_1 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_wgrad32_micro.txt
cat gpurun_out/r04_wgrad32_micro.txt
bash tools/ab_env.sh 2 SEG_WGRAD_IMPL "- old" --size 512 --steps 20 --warmup 5 --windows 3 2>&1 | tail -5
