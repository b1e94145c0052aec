cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh 3 SEG_CONV_IMPL "tiled ring" --size 512 --steps 20 --warmup 5 --windows 3 2>&1 | tail -8
cp gpurun_out/ab_env_SEG_CONV_IMPL.txt gpurun_out/r04_ab_ring_512.txt
