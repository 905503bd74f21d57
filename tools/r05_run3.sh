#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
export TMPDIR=/tmp
(timeout 600 python tools/lm_ab.py --windows 1024 --rounds 3 _ab/libssk_noflip.so srba-stereo-slam_amd/libssk_hip.so 2>&1) > gpurun_out/r05m/ab_flip.log
(timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_shared_gpu.py tests/test_exchange_gpu.py tests/test_rba_capi_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r05m/lm_tests.log
cat gpurun_out/r05m/*.log
