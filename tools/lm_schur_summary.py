#!/usr/bin/env python3
"""Summarises gpurun_out/schur/* (tools/lm_schur_profile.sh: rocprofv3 kernel trace + two PMC passes per case) into
profiles/r02_lm_mfma_vs_sparse.{json,csv}: per case the LM kernel's mean duration and its counters summed over the launch
(SQ_* in quad-cycles where the counter is a cycle count, see MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "schur")
out = {"what": "LM kernels (one workgroup per window, SSK_LM_KERNEL=tiled): Schur complement on the FP64 matrix cores (SSK_LM_SCHUR=mfma: "
               "v_mfma_f64_16x16x4_f64 on Y panels) against sparse 6x6 items (=sparse); d2 = 256 bench windows (E = 10, 1422 observations), "
               "d5 = 64 depth-5 local areas (E = 40, 8308 observations); tools/lm_schur_profile.sh", "cases": {}}
rows = []
for case in ("d2_mfma", "d2_sparse", "d5_mfma", "d5_sparse"):
    rec = {}
    db = os.path.join(SRC, case + ".trace", "t_results.db")
    c = sqlite3.connect(db)
    r = c.execute("select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), max(scratch_size) from kernels "
                  "where name like '%lm_solve%' group by name").fetchall()
    for name, n, avg, mn, mx, vg, lds, scr in r:
        import re
        rec["kernel"] = re.search(r"lm_solve_\w+", name).group(0)
        rec.update(launches=n, avg_ms=avg / 1e6, min_ms=mn / 1e6, max_ms=mx / 1e6, vgprs=vg, lds_bytes=lds, scratch_bytes_per_lane=scr)
    for p in ("pmc1", "pmc2"):
        c = sqlite3.connect(os.path.join(SRC, "%s.%s" % (case, p), "p_results.db"))
        q = c.execute("select counter_name, sum(value) / count(distinct dispatch_id) from counters_collection where kernel_name like '%lm_solve%' group by counter_name").fetchall()
        for name, v in q:
            rec[name] = v
    if rec.get("SQ_BUSY_CYCLES"):
        # VALU / MFMA / LDS pipe activity relative to the time waves were resident (quad-cycle counters summed over SIMDs)
        rec["valu_active_per_wave_cycle"] = rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_WAVE_CYCLES"]
        rec["lds_active_per_wave_cycle"] = rec["SQ_ACTIVE_INST_LDS"] / rec["SQ_WAVE_CYCLES"]
        rec["wait_any_per_wave_cycle"] = rec.get("SQ_WAIT_ANY", 0) / rec["SQ_WAVE_CYCLES"]
        rec["mfma_busy_cycles_per_launch"] = rec.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    out["cases"][case] = rec
    rows.append(dict(case=case, **rec))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_lm_mfma_vs_sparse.json"), "w"), indent=1, sort_keys=True)
keys = sorted({k for r in rows for k in r})
with open(os.path.join(ROOT, "profiles", "r02_lm_mfma_vs_sparse.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["case"] + [k for k in keys if k != "case"])
    w.writeheader()
    for r in rows:
        w.writerow(r)
for case, r in out["cases"].items():
    print(case, r.get("kernel"), "%.3f ms" % r.get("avg_ms", 0), "VALU insts %.3g" % r.get("SQ_INSTS_VALU", 0), "MFMA f64 %.3g" % r.get("SQ_INSTS_VALU_MFMA_F64", 0),
          "LDS insts %.3g" % r.get("SQ_INSTS_LDS", 0), "bank conflict %.3g" % r.get("SQ_LDS_BANK_CONFLICT", 0), "valu/wave %.3f lds/wave %.3f wait %.3f" % (r.get("valu_active_per_wave_cycle", 0), r.get("lds_active_per_wave_cycle", 0), r.get("wait_any_per_wave_cycle", 0)))
