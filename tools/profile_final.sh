#!/bin/bash
# Final profile of the session: kernel-trace statistics of the bench's sampling leg and of the training loop, small-batch traces, bench lines
export TMPDIR=/tmp
OUT=gpurun_out/prof_final
mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-train --no-cpu-baseline --no-other-configs > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_sampling.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktt -o kt -- python3 tools/train_prof.py 10 32768 > $OUT/ktt.log 2>&1
cp $(find $OUT/ktt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_train_32768.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kt -- python3 tools/small_batch.py msr3 8192 > $OUT/kts.log 2>&1
cp $(find $OUT/kts -name "*kernel_stats.csv" | head -1) $OUT/small_batch_kernel_stats_msr3_8192.csv
python3 tools/small_batch.py msr3 8192 > $OUT/small_msr3_8192.txt 2>&1
python3 tools/small_batch.py msr80 512 > $OUT/small_msr80_512.txt 2>&1
python3 tools/ab_mid.py > $OUT/mid_batches.txt 2>&1
rm -rf $OUT/kt $OUT/ktt $OUT/kts
