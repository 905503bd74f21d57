#!/usr/bin/env python3
"""The depth-5 LM batch of bench.py's `lm_depth5` object by itself (1 024 stage-2 windows, 4 distinct; one warm-up launch, two timed), for
rocprofv3 passes (tools/profile_r04.sh -> profiles/r04_lm_depth5_*).  With a directory argument: summarises a --pmc pass of FP64 / wave
counters found there into JSON on stdout instead of running."""
import csv
import ctypes as C
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summarise(src):
    acc = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            if "lm_solve" not in name and "lm_split" not in name:
                continue
            s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
    out = {"command": "rocprofv3 --kernel-trace --kernel-include-regex lm_solve --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 "
                      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 tools/lm_depth5_run.py",
           "units": "average per dispatch; instruction counters per wavefront instruction (x 64 lanes, x 2 flop for an FMA); *_CYCLES / ACTIVE / WAIT in quad-cycles "
                    "summed over SIMDs", "kernels": {}}
    for name, k in sorted(acc.items()):
        rec = {c: v[0] / v[1] for c, v in k.items()}
        rec["dispatches"] = max(v[1] for v in k.values())
        if "SQ_INSTS_VALU_FMA_F64" in rec:
            rec["executed_fp64_flops"] = 64.0 * (2.0 * rec.get("SQ_INSTS_VALU_FMA_F64", 0) + rec.get("SQ_INSTS_VALU_MUL_F64", 0) + rec.get("SQ_INSTS_VALU_ADD_F64", 0))
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in rec:             # (its own --pmc pass: the matrix cores' FP64 work, 512 flop per unit)
            rec["executed_mfma_fp64_flops"] = 512.0 * rec["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        wc = rec.get("SQ_WAVE_CYCLES", 0)
        if wc:
            rec["valu_active"] = rec.get("SQ_ACTIVE_INST_VALU", 0) / wc
            rec["waiting"] = rec.get("SQ_WAIT_ANY", 0) / wc
            if "SQ_ACTIVE_INST_LDS" in rec:
                rec["lds_active"] = rec["SQ_ACTIVE_INST_LDS"] / wc
        if rec.get("SQ_INSTS_LDS"):
            rec["lds_conflict_cycles_per_inst"] = rec.get("SQ_LDS_BANK_CONFLICT", 0) / rec["SQ_INSTS_LDS"]
        out["kernels"][name] = rec
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


def main():
    if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]):
        return summarise(sys.argv[1])
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ssk_amd import capi
    import synth_lm as sl
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ctx = pkg.Context(0)
    wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(4)]
    lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
    b = capi.LmBatch(ctx, [wins[k % 4] for k in range(n)])
    name = b.kernel_name()
    ts = []
    for rep in range(3):
        b.reset(); ctx.synchronize()
        t0 = time.perf_counter(); b.run(lp); r = b.results(); ts.append(1e3 * (time.perf_counter() - t0))
    print(json.dumps({"kernel": name, "windows": n, "launch_ms": ts[1:], "mean_iters": sum(x.iters for x in r) / len(r)}))
    b.close(); ctx.close()


if __name__ == "__main__":
    main()
