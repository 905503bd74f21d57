#!/bin/bash
# Builds _ab/libssk_<tag>.so: the in-tree objects with ONE source (csrc/<stem>.hip) recompiled under extra -D flags, for same-box A/B of a kernel's
# variants (SSK_LIB=_ab/libssk_<tag>.so python bench.py ...).  usage: bash tools/kernel_variant.sh <stem> <tag> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
stem=$1; tag=$2; shift; shift
mkdir -p _ab
src_flags=$(python -c "import sys; sys.path.insert(0, 'srba-stereo-slam_amd'); import build; print(' '.join(build.SRC_FLAGS.get('$stem.hip', [])))")
python srba-stereo-slam_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include $src_flags "$@" \
  -c srba-stereo-slam_amd/csrc/$stem.hip -o _ab/${stem}_$tag.o
objs=$(ls srba-stereo-slam_amd/csrc/_obj/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libssk_$tag.so $objs _ab/${stem}_$tag.o
echo "_ab/libssk_$tag.so"
