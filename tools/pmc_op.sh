#!/bin/bash
# Two SQ counter passes + one LDS pass over ONE operator's kernel (tools/prof_op.py), summarised.
#   tools/pmc_op.sh <op_name> <kernel substring> <out_dir>
OP=${1:-up.17.res}; PAT=${2:-k_panel128_h}; OUT=${3:-gpurun_out/pmc_op}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 tools/prof_op.py $OP > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/prof_op.py $OP > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM --output-format csv -d $OUT/p3 -- python3 tools/prof_op.py $OP > $OUT/p3.log 2>&1
tail -1 $OUT/p1.log
python3 tools/pmc_summary.py $OUT "$PAT"
