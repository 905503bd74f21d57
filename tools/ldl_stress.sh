#!/bin/bash
# Failure path of K2's matrix-core LDL^T under load (the tile waves must leave AT the failing panel): a negative pivot in every
# 16-column panel in turn (first / middle / last column of the panel), a full grid of workgroups, many solves per launch; every run must report
# ok 0 from both solvers and return (each under its own timeout: a hang is reported, not waited for).
# usage (GPU box): bash tools/ldl_stress.sh [n = 255] [workgroups = 512] [reps = 40]
cd "$(dirname "$0")/.."
n=${1:-255}; wgs=${2:-512}; reps=${3:-40}
[ -x _ab/ldl_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I srba-stereo-slam_amd/csrc tools/ldl_bench.hip -o _ab/ldl_bench || exit 1
fail=0; runs=0
for ((p = 0; 16 * p < n; ++p)); do
  for c in 0 7 15; do
    pos=$((16 * p + c)); [ $pos -lt $n ] || continue
    out=$(timeout 60 _ab/ldl_bench $n $reps $wgs $pos 2>&1); rc=$?
    runs=$((runs + 1))
    if [ $rc -ne 0 ] || ! echo "$out" | grep -q "ldl_solve_mfma_lead.*ok 0"; then echo "FAIL at pivot $pos (rc $rc): $out"; fail=$((fail + 1)); fi
  done
done
echo "ldl_stress: n $n, $wgs workgroups x $reps solves, $runs failing-pivot positions, $fail failures"
exit $fail
