#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_rba_capi_gpu.py tests/test_slam_capi_gpu.py tests/test_stream_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r05i/tests.log
(timeout 1200 python bench.py 2>gpurun_out/r05i/bench.err | tail -1) > gpurun_out/r05i/bench.json
cat gpurun_out/r05i/tests.log; tail -3 gpurun_out/r05i/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05i/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_incl_symbolic')})
print(d['stage_ms_per_launch'])
print(d['lm_depth5']['batched'], d['lm_depth5'].get('p90_window'), d['lm_depth5']['one_keyframe'])
u=d.get('unbatched') or {}
print(u.get('ms_per_keyframe_frame'), u.get('per_keyframe_ms'), u.get('lm_max_optimised_edges'))
e=d['e2e_stage2']; print(e['value'], e['ms_per_step'], e['ms'], e['define_new_keyframe_batch']); print(e['depth5']['value'], e['depth5']['ms_per_step'])
PY
