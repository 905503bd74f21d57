#!/bin/bash
# Builds _ab/libssk_<tag>.so: the in-tree objects with the phase-split LM sources (ssk_lm_split.hip, ssk_lm_split_solve.hip, ssk_capi_lm.hip) recompiled
# under extra -D flags, for same-box A/B timing (tools/lm_ab.py --depth 5).  usage: bash tools/split_variant.sh <tag> [-DSP_X=1 ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p _ab
python srba-stereo-slam_amd/build.py > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include"
for s in ssk_lm_split ssk_lm_split_solve ssk_capi_lm; do
  /opt/rocm/bin/hipcc $F "$@" -c srba-stereo-slam_amd/csrc/$s.hip -o _ab/${s}_$tag.o &
done
wait
objs=$(ls srba-stereo-slam_amd/csrc/_obj/*.o | grep -v "ssk_lm_split.o\|ssk_lm_split_solve.o\|ssk_capi_lm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libssk_$tag.so $objs _ab/ssk_lm_split_$tag.o _ab/ssk_lm_split_solve_$tag.o _ab/ssk_capi_lm_$tag.o
echo "_ab/libssk_$tag.so"
