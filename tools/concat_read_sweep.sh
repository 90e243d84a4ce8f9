#!/bin/bash
# Is the second read of the concat input (stage 1, then the Linear shortcut) of the 128-wide up block an Infinity-Cache hit?
# FETCH_SIZE, WRITE_SIZE and kernel time of up.17.res at 65 536 / 131 072 / 262 144 rows (concat input 134 / 268 / 537 MB for the two
# passes: the last is past the 256 MB cache).  usage: bash tools/concat_read_sweep.sh [out_dir]
OUT=${1:-gpurun_out/concat_sweep}
export TMPDIR=/tmp
mkdir -p $OUT
for B in 65536 131072 262144; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$B -- python3 tools/prof_op.py up.17.res $B 10 > $OUT/f$B.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w$B -- python3 tools/prof_op.py up.17.res $B 10 > $OUT/w$B.log 2>&1
  python3 tools/prof_op.py up.17.res $B 20 > $OUT/t$B.log 2>&1
  echo "== B=$B"; tail -1 $OUT/t$B.log
  python3 tools/pmc_summary.py $OUT/f$B "k_panel128_h<true, 0, 1, 2>"; python3 tools/pmc_summary.py $OUT/w$B "k_panel128_h<true, 0, 1, 2>"
done > $OUT/summary.txt 2>&1
rm -rf $OUT/f65536 $OUT/f131072 $OUT/f262144 $OUT/w65536 $OUT/w131072 $OUT/w262144
