#!/bin/bash
# Evidence for DESIGN.md 5 "MFMA": the Schur complement of the LM kernels on the FP64 matrix cores (SSK_LM_SCHUR=mfma) against the
# sparse 6x6-item form (=sparse), at E = 10 (the bench window, 256 per launch) and E = 40 (depth-5 local area, 64 per launch).
# Run on the GPU box from the repo root: bash tools/lm_schur_profile.sh ; writes gpurun_out/schur/*, summarised by
# tools/lm_schur_summary.py into profiles/r02_lm_mfma_vs_sparse.{csv,json}.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/schur
rm -rf $OUT; mkdir -p $OUT
for shape in "2 256" "5 64"; do
  set -- $shape; depth=$1; n=$2
  for m in mfma sparse; do
    tag=d${depth}_$m
    export SSK_LM_SCHUR=$m LM_DEPTH=$depth SSK_LM_KERNEL=tiled
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/$tag.trace -o t -- python3 $OLDPWD/tools/lm_stamps.py $n > $OUT/$tag.trace.log 2>&1)
    (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $OUT/$tag.pmc1 -o p -- python3 $OLDPWD/tools/lm_stamps.py $n > $OUT/$tag.pmc1.log 2>&1)
    (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 -d $OUT/$tag.pmc2 -o p -- python3 $OLDPWD/tools/lm_stamps.py $n > $OUT/$tag.pmc2.log 2>&1)
  done
done
find $OUT -name "*.csv" | head -40
