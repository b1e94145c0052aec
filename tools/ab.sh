#!/bin/bash
# On the GPU box: same-box A/B of two prebuilt libraries (box-to-box spread is +-2 %): segmentation_amd/libseg_hip_old.so vs libseg_hip_new.so; args are passed to bench.py
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['hip_graph'])"; }
for r in 1 2 3; do
  cp segmentation_amd/libseg_hip_old.so segmentation_amd/libseg_hip.so; touch segmentation_amd/libseg_hip.so; one old "$1"
  cp segmentation_amd/libseg_hip_new.so segmentation_amd/libseg_hip.so; touch segmentation_amd/libseg_hip.so; one new "$1"
done
