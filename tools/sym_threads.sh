for th in 64 32 16 8; do
  SSK_LM_BUILD_THREADS=$th python bench.py --steps 8 --warmup 2 --unbatched-frames 0 --lm-depth5 0 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 2>/dev/null | tail -1 > gpurun_out/sym_$th.json
  python - <<PY
import json
d=json.load(open("gpurun_out/sym_$th.json")); l=d["lm_symbolic"]
print($th, round(d["value"]), round(d["value_incl_symbolic"]), round(l["lm_batch_create_ms"],2), round(l["lm_symbolic_ms_per_batch"],2), l["host_threads"])
PY
done
