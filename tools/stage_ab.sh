#!/bin/bash
# same-box A/B of library variants on bench.py's per-stage times: usage: bash tools/stage_ab.sh <stage key> lib_a.so lib_b.so ...
key=$1; shift
for lib in "$@"; do
  SSK_LIB=$lib python bench.py --steps 10 --warmup 2 --unbatched-frames 0 --lm-depth5 0 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --e2e-streams 0 2>/dev/null | tail -1 > gpurun_out/ab.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab.json"))
print("$lib", round(d["value"]), round(d["stage_ms_per_launch"]["$key"], 3))
PY
done
