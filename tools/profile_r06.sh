#!/bin/bash
# Round-6 profile artefacts (run on the GPU box from the repo root: SSK_GIT_HEAD=<head> bash tools/profile_r06.sh), summarised into gpurun_out/r06prof/
# and copied to profiles/ by hand:
#   r06_kernel_stats_bench_b1024.csv   rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time)
#   r06_pmc_traffic.json               separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.py), stamped with the kernels' durations + git head
#   r06_pmc_sq.json                    one --pmc pass of SQ counters: VALU / LDS activity of every kernel (tools/pmc_sq.py)
#   r06_pmc_lm_wave_time.json          where the LM node kernel's wave time goes (tools/pmc_lm.sh: four --pmc passes)
#   r06_lm_depth5_kernel_stats.csv / r06_lm_depth5_pmc.json / r06_lm_depth5_pmc_lds.json / r06_lm_depth5_pmc_mfma.json / r06_lm_depth5_traffic.json
#                                      the depth-5 LM batch (tools/lm_depth5_run.py, the phase-split kernels): --stats pass, FP64 / wave counters, LDS
#                                      counters, FETCH / WRITE sizes
set -u
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/r06prof
rm -rf $OUT; mkdir -p $OUT
B="--steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-streams 0"
BS="--steps 3 --warmup 1 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-streams 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOTD/bench.py $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOTD/bench.py $BS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOTD/bench.py $BS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o q -- python3 $ROOTD/bench.py $BS > $OUT/sq.log 2>&1
cd $ROOTD
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r06_pmc_traffic.json $OUT/stats
python3 tools/pmc_sq.py $OUT/pmc_sq $OUT/r06_pmc_sq.json > /dev/null 2>&1 || true
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r06_kernel_stats_bench_b1024.csv
tail -1 $OUT/stats.log | cut -c1-300
bash tools/pmc_lm.sh r06 > $OUT/pmc_lm.log 2>&1
cp gpurun_out/pmc_lm_r06/summary.json $OUT/r06_pmc_lm_wave_time.json
# the depth-5 LM batch (bench.py's lm_depth5 object): per-kernel time, executed FP64 work, LDS conflicts, HBM traffic
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d5_stats -o s -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "lm_split" --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/d5_pmc -o p -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "lm_split" --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/d5_pmc2 -o p -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "lm_split" --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/d5_pmc3 -o p -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "lm_split" --pmc FETCH_SIZE --output-format csv -d $OUT/d5_fetch -o f -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "lm_split" --pmc WRITE_SIZE --output-format csv -d $OUT/d5_write -o w -- python3 $ROOTD/tools/lm_depth5_run.py > $OUT/d5_write.log 2>&1
cd $ROOTD
cp $(find $OUT/d5_stats -name "*kernel_stats.csv" | head -1) $OUT/r06_lm_depth5_kernel_stats.csv
python3 tools/lm_depth5_run.py $OUT/d5_pmc > $OUT/r06_lm_depth5_pmc.json
python3 tools/lm_depth5_run.py $OUT/d5_pmc2 > $OUT/r06_lm_depth5_pmc_lds.json
python3 tools/lm_depth5_run.py $OUT/d5_pmc3 > $OUT/r06_lm_depth5_pmc_mfma.json
python3 tools/pmc_traffic.py $OUT/d5_fetch $OUT/d5_write $OUT/r06_lm_depth5_traffic.json $OUT/d5_stats
tail -1 $OUT/d5_stats.log | cut -c1-300
ls -la $OUT/*.json $OUT/*.csv
