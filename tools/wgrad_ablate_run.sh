#!/bin/bash
# stand-alone filter-gradient timings of the ablation arms (tools/wgrad_ablate.sh built them)
L=${1:-conv3_2,conv2_2,conv7_1,conv4_2}
echo "== full kernel"; python tools/wgrad_micro.py --size 512 --cfgs 0 --ks 0 --layers $L 2>/dev/null | cut -c1-150
for b in 1 2 4 8; do
  echo "== without bit $b (1 MFMAs, 2 LDS fragment reads, 4 LDS commits, 8 global loads)"
  SEG_LIB_PATH=$PWD/segmentation_amd/build/libseg_wabl_$b.so python tools/wgrad_micro.py --size 512 --cfgs 0 --ks 0 --layers $L 2>/dev/null | cut -c1-150
done
