#!/bin/bash
# Round profile: kernel-trace stats of the bench command + PMC passes (separate runs), summaries under gpurun_out/prof_round/
# usage (on the GPU box): bash tools/profile_round.sh
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-train --no-cpu-baseline --no-other-configs > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_sampling.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktt -o kt -- python3 tools/train_prof.py 10 32768 > $OUT/ktt.log 2>&1
cp $(find $OUT/ktt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_train_32768.csv
bash tools/pmc_bench.sh $OUT/pmc
KERNELS=("k_panel128_h<true, 0, 1, 2>" "k_panel128_h<false, 0, 1, 2>" "k_panel128_h<false, 1, 2, 2>" "k_panel128_h<true, 2, 3, 2>" "k_res64_lds<true, 0>" "k_res64_lds<true, 4>" "k_res64_dual" "k_fused_narrow_lds<2>" "k_linear_h<4, 1, 0, false>" "k_update")
for k in "${KERNELS[@]}"; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT/pmc "$k"
done > $OUT/pmc_summary.txt
# launches per reverse step of each kernel: the step-level traffic figure of the bench line comes from here
python3 tools/make_traffic.py $OUT/pmc_summary.txt "k_panel128_h<true, 0, 1, 2>" $OUT/traffic.json "profiles/${ROUND_TAG:-r04}_pmc_summary.txt" \
  "k_panel128_h<true, 0, 1, 2>=2" "k_panel128_h<false, 0, 1, 2>=1" "k_panel128_h<false, 1, 2, 2>=1" "k_panel128_h<true, 2, 3, 2>=1" "k_res64_lds<true, 0>=2" \
  "k_res64_lds<true, 4>=1" "k_res64_dual=1" "k_fused_narrow_lds<2>=2" "k_linear_h<4, 1, 0, false>=1" "k_update=1" > /dev/null
rm -rf $OUT/kt $OUT/ktt $OUT/pmc/p1 $OUT/pmc/p2 $OUT/pmc/p3 $OUT/pmc/p4 $OUT/pmc/p5 $OUT/pmc/p6
