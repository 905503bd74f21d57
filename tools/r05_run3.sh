#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r05c/gpu_tests.log
(timeout 600 python bench.py --steps 10 --warmup 3 2>gpurun_out/r05c/bench.err | tail -1) > gpurun_out/r05c/bench.json
cat gpurun_out/r05c/gpu_tests.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','roofline')})
print(d.get('stage_ms_per_launch'))
print(d.get('lm_depth5',{}).get('batched'))
print(d.get('unbatched',{}).get('ms_per_keyframe_frame'), d.get('unbatched',{}).get('per_keyframe_ms'))
print(d.get('e2e'))
PY
