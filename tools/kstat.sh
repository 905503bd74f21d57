#!/bin/bash
# per-kernel time of the default bench command (run on the GPU box from the repo root): rocprofv3 --kernel-trace --stats, top rows printed
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/kstat
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $ROOTD/bench.py --steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-streams 0 > $OUT/log 2>&1
cd $ROOTD
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < ${1:-14}: print("%-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 $OUT/log | cut -c1-200
