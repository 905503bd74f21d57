#!/bin/bash
# Builds _ab/libssk_<tag>.so: the in-tree objects with ssk_lm_node.hip recompiled under extra -D flags (same-box A/B timing of LM kernel
# variants: tools/lm_ab.py).  usage: bash tools/lm_variant.sh <tag> [-DND_PIPE=1 ...]   (run in the build container; _ab/ travels with gpurun)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p _ab
python srba-stereo-slam_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include "$@" \
  -c srba-stereo-slam_amd/csrc/ssk_lm_node.hip -o _ab/node_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include "$@" \
  -c srba-stereo-slam_amd/csrc/ssk_capi_lm.hip -o _ab/capi_lm_$tag.o
objs=$(ls srba-stereo-slam_amd/csrc/_obj/*.o | grep -v "ssk_lm_node.o\|ssk_capi_lm.o")
extra=""
if [ -n "$LM_VARIANT_TILED" ]; then     # (the tiled / cooperative kernels under the same flags too)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include "$@" \
    -c srba-stereo-slam_amd/csrc/ssk_lm_tiled.hip -o _ab/tiled_$tag.o
  objs=$(echo "$objs" | grep -v "ssk_lm_tiled.o"); extra=_ab/tiled_$tag.o
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libssk_$tag.so $objs _ab/node_$tag.o _ab/capi_lm_$tag.o $extra
echo "_ab/libssk_$tag.so"
