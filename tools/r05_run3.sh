#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
export TMPDIR=/tmp
(SSK_STREAM_HIST=gpurun_out/r05h/stream_hist.json timeout 600 python tools/stream_full.py 4541 160 2>&1 | tail -6) > gpurun_out/r05h/stream.log
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r05h/gpu_tests.log
(timeout 900 python bench.py 2>gpurun_out/r05h/bench.err | tail -1) > gpurun_out/r05h/bench.json
cat gpurun_out/r05h/stream.log gpurun_out/r05h/gpu_tests.log; tail -3 gpurun_out/r05h/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05h/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_incl_symbolic')})
print(d['stage_ms_per_launch'])
print(d['lm_depth5']['batched'] if d.get('lm_depth5') else None)
u=d.get('unbatched') or {}
print(u.get('ms_per_keyframe_frame'), u.get('per_keyframe_ms'), u.get('lm_max_optimised_edges'))
e=d['e2e_stage2']; print(e['value'], e['ms_per_step'], e['ms'], e['define_new_keyframe_batch']); print(e['depth5']['value'], e['depth5']['ms_per_step'])
PY
