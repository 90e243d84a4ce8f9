#!/bin/bash
# Exact-float32 path: kernel-trace statistics and two PMC passes over tools/f32_trace.py (65 536 rows, T = 20), summaries under gpurun_out/prof_f32/
# usage (on the GPU box): bash tools/profile_f32.sh
export TMPDIR=/tmp
OUT=gpurun_out/prof_f32
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 tools/f32_trace.py 65536 3 > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/f32_exact_kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 tools/f32_trace.py 65536 1 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/f32_trace.py 65536 1 > $OUT/p2.log 2>&1
for k in "k_resblock<128, true, true>" "k_resblock_lin<128, true, 3, true>" "k_resblock<128, false, true>" "k_resblock<64, true, true>" "k_fused_narrow("; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT "$k"
done > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/kt $OUT/p1 $OUT/p2
