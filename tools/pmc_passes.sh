#!/bin/bash
# rocprofv3 PMC passes (counters only, no tracing) for one operator kernel; run on the GPU box from the repo root.
#   tools/pmc_passes.sh <op_name> <out_dir>
OP=${1:-up.17.res}; OUT=${2:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/p1 -- python3 tools/prof_op.py $OP > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/prof_op.py $OP > $OUT/p2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCC_HIT TCC_MISS TCC_REQ TA_TA_BUSY --output-format csv -d $OUT/p3 -- python3 tools/prof_op.py $OP > $OUT/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- python3 tools/prof_op.py $OP > $OUT/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -- python3 tools/prof_op.py $OP > $OUT/p5.log 2>&1
tail -2 $OUT/p1.log
