#!/bin/bash
# PMC passes over a few training steps (32 768 rows); summarise with tools/pmc_summary.py <dir> <kernel>
OUT=${1:-gpurun_out/pmc_train}
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="tools/train_prof.py 4 32768"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p5 -- python3 $ARGS > $OUT/p5.log 2>&1
for k in "k_resblock_bwd_h<128, true>" "k_resblock_bwd_h<128, false>" "k_resblock_bwd_h<64, true>" "k_wgrad_h" "k_wide128_h<true, 0" "k_fused_narrow_h"; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT "$k"
done > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p5
