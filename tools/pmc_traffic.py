#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate passes, as the TCC block cannot hold both) of
`python3 bench.py --steps 3 --warmup 1 --cpu-sample 0` into profiles/r02_pmc_traffic.json (per-kernel averages, KB per
dispatch).  Usage on the GPU box, from the repo root:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json
bench.py turns the numbers into bytes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KB units; FETCH_SIZE tallies half the bytes of
a wide coalesced read on gfx950, MI355X_MICROARCH.md; re-checked here on lm_reset_kernel, which reads what it writes)."""
import csv
import glob
import json
import os
import re
import sys


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            if "kernel" not in name or "at::" in name:
                continue
            s = out.setdefault(name, [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
    return out


def durations(d):
    """average duration (ms) per kernel from the kernel trace of a pass"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            s = out.setdefault(name, [0.0, 0])
            s[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6; s[1] += 1
    return {k: v[0] / v[1] for k, v in out.items()}


def main():
    fd, wd, dst = sys.argv[1:4]
    stats_dir = sys.argv[4] if len(sys.argv) > 4 else None      # an un-counted kernel-trace pass of the same command: its durations are the ones the bench must reproduce
    fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    dur = durations(stats_dir) if stats_dir else durations(fd)
    kernels = {}
    for k in sorted(set(fe) & set(wr)):
        kernels[k] = {"fetch_size_kb": round(fe[k][0] / fe[k][1], 2), "write_size_kb": round(wr[k][0] / wr[k][1], 2), "dispatches": fe[k][1],
                      "avg_ms": round(dur[k], 5) if k in dur else None}
    cal = kernels.get("lm_reset_kernel")
    out = {"batch": int(os.environ.get("SSK_PMC_BATCH", "1024")),
           "git_head": os.environ.get("SSK_GIT_HEAD"),
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 (two separate passes), aggregated by tools/pmc_traffic.py",
           "units": "KB per dispatch, average over the dispatches of the run; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950",
           "calibration": "lm_reset_kernel reads exactly the bytes it writes (coalesced 8 B/lane copy): FETCH_SIZE/WRITE_SIZE = %.3f, i.e. FETCH_SIZE tallies half the bytes, as MI355X_MICROARCH.md states for gfx950"
                          % (cal["fetch_size_kb"] / cal["write_size_kb"] if cal else float("nan")),
           "kernels": kernels}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        print("%-28s %10.1f MB per dispatch" % (k, (2 * v["fetch_size_kb"] + v["write_size_kb"]) / 1024))


if __name__ == "__main__":
    main()
