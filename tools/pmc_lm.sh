#!/bin/bash
# Where the LM kernel's wave time goes: three --pmc passes over the default bench command, LM kernels only
# (run on the GPU box from the repo root: bash tools/pmc_lm.sh [tag]); summarised by tools/pmc_lm.py into profiles/.
set -u
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/pmc_lm${1:+_$1}
rm -rf $OUT; mkdir -p $OUT
BS="--steps 3 --warmup 1 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-streams 0"
cd /tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SALU" \
  "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_FLAT SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --kernel-include-regex "lm_solve" --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $ROOTD/bench.py $BS > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log | cut -c1-200
done
cd $ROOTD
python3 tools/pmc_lm.py $OUT $OUT/summary.json
