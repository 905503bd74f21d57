#!/bin/bash
# Builds _ab/libssk_<tag>.so: the in-tree objects with ssk_describe.hip recompiled under extra -D flags (same-box A/B of describe_kernel variants:
# SSK_LIB=_ab/libssk_<tag>.so python bench.py ...).  usage: bash tools/desc_variant.sh <tag> [-DSSK_DESC_HT=50 ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p _ab
python srba-stereo-slam_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include "$@" \
  -c srba-stereo-slam_amd/csrc/ssk_describe.hip -o _ab/describe_$tag.o
objs=$(ls srba-stereo-slam_amd/csrc/_obj/*.o | grep -v "ssk_describe.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libssk_$tag.so $objs _ab/describe_$tag.o
echo "_ab/libssk_$tag.so"
