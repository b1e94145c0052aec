#!/bin/bash
# A/B on one box: builds the library of a git revision (default HEAD) into segmentation_amd/build/libseg_<tag>.so next to the
# working-tree build; run an arm with SEG_LIB_PATH=<that file>.   tools/ab_build.sh [rev] [tag]
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; tag=${2:-base}
d=segmentation_amd/build/ab_$tag; rm -rf $d; mkdir -p $d/pkg/csrc $d/include
for f in $(git ls-tree --name-only $rev segmentation_amd/csrc/); do git show $rev:$f > $d/pkg/csrc/$(basename $f); done
git show $rev:include/seg_hip.h > $d/include/seg_hip.h
objs=""
for src in $d/pkg/csrc/*.hip; do            # whatever translation units that revision has
  f=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$d/include -c $src -o $d/$f.o &
  objs="$objs $d/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o segmentation_amd/build/libseg_$tag.so $objs
echo built segmentation_amd/build/libseg_$tag.so
