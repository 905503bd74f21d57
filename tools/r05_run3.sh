#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
export TMPDIR=/tmp
(for n in 240 252 264 330 372 396; do timeout 60 _ab/ldl_bench $n 50 1 | head -2; done; timeout 120 _ab/ldl_bench 372 20 256 | head -2; timeout 60 _ab/ldl_bench 372 5 64 300 | head -2; timeout 60 _ab/ldl_bench 372 5 64 50 | head -2) > gpurun_out/r05g/ldl_bench.log 2>&1
(timeout 600 bash tools/ldl_stress.sh 255 512 20 2>&1 | tail -3) > gpurun_out/r05g/ldl_stress.log
(timeout 900 python -m pytest tests/test_lm_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r05g/lm_tests.log
(SSK_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --unbatched-frames 0 --lm-depth5 0 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --e2e-depth5-streams 0 2>&1 | grep -i "lean\|create\|ssk lm" | tail -30) > gpurun_out/r05g/timing.log
cat gpurun_out/r05g/*.log
