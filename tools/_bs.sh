#!/bin/bash
timeout 600 python -m pytest tests/test_lm_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python tools/lm_ab.py --depth 8 --windows 64 --rounds 2 _ab/libssk_prev.so srba-stereo-slam_amd/libssk_hip.so
timeout 300 python3 tools/lm_lead_stress.py 2>&1 | tail -1
