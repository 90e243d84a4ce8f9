#!/bin/bash
# Round-6 profile set (GPU box): bench lines, kernel-trace statistics (sampling, training, small batches), PMC passes of the reverse step and of the
# training step (counters only, FETCH_SIZE / WRITE_SIZE in passes of their own), per-kernel summaries, traffic.json / traffic_train.json.
# usage: bash tools/profile_r06.sh   -> gpurun_out/prof_r06/
export TMPDIR=/tmp
OUT=gpurun_out/prof_r06
mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-train --no-cpu-baseline --no-other-configs > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_sampling.csv
DEVICE_DRAWS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktt -o kt -- python3 tools/train_prof.py 10 32768 > $OUT/ktt.log 2>&1
cp $(find $OUT/ktt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_train_32768.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kt -- python3 tools/small_batch.py msr3 8192 > $OUT/kts.log 2>&1
cp $(find $OUT/kts -name "*kernel_stats.csv" | head -1) $OUT/small_batch_kernel_stats_msr3_8192.csv
python3 tools/small_batch.py msr3 8192 > $OUT/small_msr3_8192.txt 2>&1
python3 tools/small_batch.py msr80 512 > $OUT/small_msr80_512.txt 2>&1
python3 tools/ab_mid.py > $OUT/mid_batches.txt 2>&1
python3 tools/train_host_prof.py 32768 > $OUT/train_host_prof.txt 2>&1
bash tools/pmc_bench.sh $OUT/pmc
KERNELS=("k_panel128_h<true, 0, 1, 2>" "k_panel128_h<false, 0, 1, 2>" "k_panel128_h<false, 1, 2, 2>" "k_panel128_h<true, 2, 3, 2>" "k_res64_lds<true, 0>" "k_res64_lds<true, 4>" "k_res64_dual" "k_fused_narrow_lds<2>" "k_linear_h<4, 1, 0, false>" "k_update")
for k in "${KERNELS[@]}"; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT/pmc "$k"
done > $OUT/pmc_summary.txt
python3 tools/make_traffic.py $OUT/pmc_summary.txt "k_panel128_h<true, 0, 1, 2>" $OUT/traffic.json "profiles/r06_pmc_summary.txt" \
  "k_panel128_h<true, 0, 1, 2>=2" "k_panel128_h<false, 0, 1, 2>=1" "k_panel128_h<false, 1, 2, 2>=1" "k_panel128_h<true, 2, 3, 2>=1" "k_res64_lds<true, 0>=2" \
  "k_res64_lds<true, 4>=1" "k_res64_dual=1" "k_fused_narrow_lds<2>=1" "k_linear_h<4, 1, 0, false>=1" "k_update=1" > /dev/null
bash tools/pmc_train.sh $OUT/pmc_train
cp $OUT/pmc_train/summary.txt $OUT/pmc_train_summary.txt
rm -rf $OUT/kt $OUT/ktt $OUT/kts $OUT/pmc/p1 $OUT/pmc/p2 $OUT/pmc/p3 $OUT/pmc/p4 $OUT/pmc/p5 $OUT/pmc/p6
