#!/bin/bash
# SQ counters of ONE kernel (regex $1, default fast_nms) on the default bench command (run on the GPU box from the repo root) -> gpurun_out/pmc_fast/summary
export TMPDIR=/tmp
RE=${1:-fast_nms}
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/pmc_fast
rm -rf $OUT; mkdir -p $OUT
BS="--steps 3 --warmup 1 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-streams 0"
cd /tmp
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/a -o q -- python3 $ROOTD/bench.py $BS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o q -- python3 $ROOTD/bench.py $BS > $OUT/b.log 2>&1
cd $ROOTD
python3 tools/pmc_sq.py $OUT/a $OUT/a.json
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$OUT/b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s = acc.setdefault(r["Counter_Name"], [0.0, 0]); s[0] += float(r["Counter_Value"]); s[1] += 1
print({k: v[0] / v[1] for k, v in acc.items()})
PY
