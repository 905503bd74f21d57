#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_fmat_gpu.py tests/test_frontend_gpu.py -x -q 2>&1 | tail -12) > gpurun_out/r05e/tests.log
(bash tools/stage_ab.sh fund_matrix srba-stereo-slam_amd/libssk_hip.so _ab/libssk_fm128.so _ab/libssk_fm128w3.so _ab/libssk_fm64.so 2>&1) > gpurun_out/r05e/fm_ab.log
cat gpurun_out/r05e/*.log
