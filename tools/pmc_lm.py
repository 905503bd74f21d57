#!/usr/bin/env python3
"""Summary of tools/pmc_lm.sh: per LM kernel, every collected SQ counter averaged per dispatch, and the shares of wave time.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_INST_LEVEL_x / SQ_INSTS_x = mean latency of
an x instruction in cycles (MI355X_MICROARCH.md, 'rocprofv3 PMC slots')."""
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
            s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
    out = {}
    for name, k in sorted(acc.items()):
        rec = {c: v[0] / v[1] for c, v in k.items()}
        wc = rec.get("SQ_WAVE_CYCLES", 0)
        d = {}
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if c in rec:
                    d["share_" + c[3:].lower()] = rec[c] / wc
        for x in ("VMEM", "LDS", "SMEM"):
            n = rec.get("SQ_INSTS_" + x, 0) or (rec.get("SQ_INSTS_VMEM_RD", 0) + rec.get("SQ_INSTS_VMEM_WR", 0) if x == "VMEM" else 0)
            if n and "SQ_INST_LEVEL_" + x in rec:
                d["latency_cycles_" + x.lower()] = rec["SQ_INST_LEVEL_" + x] / n
        f64 = sum(rec.get("SQ_INSTS_VALU_%s_F64" % t, 0) for t in ("FMA", "MUL", "ADD", "TRANS"))
        if f64 and rec.get("SQ_INSTS_VALU"):
            d["f64_share_of_valu_insts"] = f64 / rec["SQ_INSTS_VALU"]
            d["f64_flop"] = 64.0 * (2 * rec.get("SQ_INSTS_VALU_FMA_F64", 0) + rec.get("SQ_INSTS_VALU_MUL_F64", 0) + rec.get("SQ_INSTS_VALU_ADD_F64", 0))
        rec["derived"] = d
        out[name] = rec
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for name, r in out.items():
        print(name)
        for c, v in sorted(r.items()):
            if c != "derived":
                print("   %-28s %.4g" % (c, v))
        for c, v in sorted(r["derived"].items()):
            print("   -> %-34s %.4g" % (c, v))


if __name__ == "__main__":
    main()
