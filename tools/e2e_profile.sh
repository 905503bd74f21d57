#!/bin/bash
# e2e_stage2 alone: per-step host phases (SSK_TIMING: the creates' breakdown on stderr) and, second run, the kernel trace of the same pass.
# usage (GPU box): bash tools/e2e_profile.sh <tag>
tag=${1:-e2e}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
ARGS="--steps 3 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-depth5-streams 0"
SSK_BENCH_E2E_STEPS=1 SSK_TIMING=1 python3 bench.py $ARGS > $out/line.json 2> $out/timing.err
python3 - <<PY
import json
d=json.loads(open("$out/line.json").read().strip().splitlines()[-1])
e=d["e2e_stage2"]
print("e2e_stage2 value", e["value"], "ms_per_step", e["ms_per_step"], "groups", e["stream_groups"], "group0", e["group0_ms_per_step"], e["group0_ms"])
if "serial" in e: print("serial value", e["serial"]["value"], "ms_per_step", e["serial"]["ms_per_step"], e["serial"]["group0_ms"], {k: v for k, v in e["serial"]["define_new_keyframe_batch"].items() if k.endswith("_ms")})
for s in e["define_new_keyframe_batch"].get("per_step", []):
    print({k:v for k,v in s.items() if not k.endswith("kernel")})
print(e["define_new_keyframe_batch"]["stage2_kernel"])
PY
grep "ssk timing" $out/timing.err | tail -24
