#!/bin/bash
# Round-2 profile artefacts (run on the GPU box from the repo root: bash tools/profile_r02.sh), summarised into profiles/:
#   r02_kernel_stats_bench_b1024.csv   rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time)
#   r02_pmc_traffic.json               separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.py)
#   r02_pmc_sq.json                    one --pmc pass of SQ counters: VALU / LDS activity of every kernel (tools/pmc_sq.py)
set -u
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/r02prof
rm -rf $OUT; mkdir -p $OUT
B="--steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0"
BS="--steps 3 --warmup 1 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOTD/bench.py $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOTD/bench.py $BS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOTD/bench.py $BS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o q -- python3 $ROOTD/bench.py $BS > $OUT/sq.log 2>&1
cd $ROOTD
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
tail -2 $OUT/stats.log | cut -c1-300
