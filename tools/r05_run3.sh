#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_rba_capi_gpu.py -x -q 2>&1 | tail -6) > gpurun_out/r05j/tests.log
cat gpurun_out/r05j/tests.log
SSK_GIT_HEAD=ae277c4 timeout 2400 bash tools/profile_r05.sh > gpurun_out/r05j/profile.log 2>&1
tail -30 gpurun_out/r05j/profile.log
(timeout 1200 python bench.py 2>gpurun_out/r05j/bench.err | tail -1) > gpurun_out/r05j/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05j/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_incl_symbolic')})
print(d['roofline'])
print(d['stage_ms_per_launch'])
print(d['lm_depth5']['batched'], d['lm_depth5'].get('p90_window'))
u=d.get('unbatched') or {}
print(u.get('ms_per_keyframe_frame'), u.get('per_keyframe_ms'))
e=d['e2e_stage2']; print(e['value'], e['ms_per_step'], e['ms']); print(e['depth5']['value'], e['depth5']['ms_per_step'])
PY
