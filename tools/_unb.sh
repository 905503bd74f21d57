#!/bin/bash
mkdir -p gpurun_out/r05w
SSK_TIMING=1 timeout 600 python3 tools/stream_latency.py loop 161 > gpurun_out/r05w/stream.log 2> gpurun_out/r05w/stream.err
tail -3 gpurun_out/r05w/stream.log
grep "lean (1 windows" gpurun_out/r05w/stream.err | tail -40 | cut -c1-330
grep -c "lean (1 windows" gpurun_out/r05w/stream.err
