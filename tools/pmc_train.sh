#!/bin/bash
# PMC passes over a few training steps (32 768 rows), counters only, FETCH_SIZE / WRITE_SIZE in passes of their own;
# summarise with tools/pmc_summary.py <dir> <kernel>.   usage: bash tools/pmc_train.sh [out_dir]
OUT=${1:-gpurun_out/pmc_train}
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="tools/train_prof.py 4 32768"
export DEVICE_DRAWS=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- python3 $ARGS > $OUT/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -- python3 $ARGS > $OUT/p5.log 2>&1
for k in "k_wgrad_h" "k_resblock_bwd_c<128, true>" "k_resblock_bwd_c<128, false>" "k_resblock_bwd_c<64, true>" "k_resblock_bwd_c<64, false>" "k_fused_narrow_bwd_h" "k_fused_narrow_h" "k_resblock_c<128, true>" "k_resblock_c<128, false>" "k_colsum" "k_reduce_ranges"; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT "$k"
done > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
