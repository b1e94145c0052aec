#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_arms.sh 3 ws_256 "SEG_WGRAD_STREAMS=2" "SEG_WGRAD_STREAMS=3" "SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=48" -- --size 256 || exit 1
bash tools/ab_arms.sh 2 ws_512 "SEG_WGRAD_STREAMS=2" "SEG_WGRAD_STREAMS=3" "SEG_WGRAD_STREAMS=3 SEG_WGRAD_WGS=96" -- --size 512 || exit 1
