#!/bin/bash
# bench.py's default line with the LM solves on the front-end's stream (--overlap 0, the default) and on a second stream (--overlap 1), twice each
for o in 0 1 0 1; do
  python bench.py --overlap $o --steps 20 --warmup 3 --unbatched-frames 0 --lm-depth5 0 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 2>/dev/null | tail -1 > gpurun_out/o.json
  python - <<PY
import json
d = json.load(open("gpurun_out/o.json"))
print("overlap", $o, round(d["value"]), round(d["ms_per_step"], 3))
PY
done
