#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass of SQ counters (tools/profile_r02.sh) -> profiles/r02_pmc_sq.json.
SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* are quad-cycles summed over the SIMDs (MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import csv
import glob
import json
import os
import re
import sys


def main():
    src, dst = sys.argv[1:3]
    acc = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            if "kernel" not in name or "at::" in name:
                continue
            k = acc.setdefault(name, {})
            s = k.setdefault(r["Counter_Name"], [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
    out = {"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY "
                      "--output-format csv -- python3 bench.py --steps 3 --warmup 1 (cpu / pcie / un-batched passes off)",
           "units": "average per dispatch; *_CYCLES, ACTIVE_INST_*, WAIT_* in quad-cycles summed over SIMDs; derived: valu_active = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES "
                    "(share of resident-wave time issuing VALU), lds_active likewise, waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
           "kernels": {}}
    for name, k in sorted(acc.items()):
        rec = {c: v[0] / v[1] for c, v in k.items()}
        rec["dispatches"] = max(v[1] for v in k.values())
        wc = rec.get("SQ_WAVE_CYCLES", 0)
        if wc:
            rec["valu_active"] = rec.get("SQ_ACTIVE_INST_VALU", 0) / wc
            rec["lds_active"] = rec.get("SQ_ACTIVE_INST_LDS", 0) / wc
            rec["waiting"] = rec.get("SQ_WAIT_ANY", 0) / wc
        out["kernels"][name] = rec
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for name, r in out["kernels"].items():
        if "valu_active" in r:
            print("%-28s valu %.3f lds %.3f waiting %.3f  VALU insts %.3g LDS insts %.3g bank conflicts %.3g" % (
                name, r["valu_active"], r["lds_active"], r["waiting"], r.get("SQ_INSTS_VALU", 0), r.get("SQ_INSTS_LDS", 0), r.get("SQ_LDS_BANK_CONFLICT", 0)))


if __name__ == "__main__":
    main()
