#!/bin/bash
# Builds the ablation arms of the filter-gradient kernel (-DSEG_WABL=bits, see conv_wgrad.hip) into
# segmentation_amd/build/libseg_wabl_<bits>.so; run an arm with SEG_LIB_PATH=<that file> python tools/wgrad_micro.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p segmentation_amd/build/abl
for bits in "$@"; do
  d=segmentation_amd/build/abl/$bits; mkdir -p $d
  for f in conv_fwd conv_first elementwise deconv_ops adv_ops; do
    [ -f segmentation_amd/build/$f.o ] || { echo "build the library first"; exit 1; }
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DSEG_WABL=$bits -Iinclude -c segmentation_amd/csrc/conv_wgrad.hip -o $d/conv_wgrad.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o segmentation_amd/build/libseg_wabl_$bits.so $d/conv_wgrad.o \
     segmentation_amd/build/conv_fwd.o segmentation_amd/build/conv_first.o segmentation_amd/build/elementwise.o segmentation_amd/build/deconv_ops.o segmentation_amd/build/adv_ops.o
  echo built segmentation_amd/build/libseg_wabl_$bits.so
done
