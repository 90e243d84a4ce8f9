OUT=gpurun_out/prof_round; mkdir -p $OUT
bash tools/pmc_bench.sh $OUT/pmc
KERNELS=("k_panel128_h<true, 0, 1, 2>" "k_panel128_h<false, 0, 1, 2>" "k_panel128_h<false, 1, 2, 2>" "k_panel128_h<true, 2, 3, 2>" "k_res64_lds<true, 0>" "k_res64_lds<true, 4>" "k_res64_dual" "k_fused_narrow_lds<2>" "k_linear_h<4, 1, 0, false>" "k_update")
for k in "${KERNELS[@]}"; do echo "== $k"; python3 tools/pmc_summary.py $OUT/pmc "$k"; done > $OUT/pmc_summary.txt
python3 tools/make_traffic.py $OUT/pmc_summary.txt "k_panel128_h<true, 0, 1, 2>" $OUT/traffic.json "profiles/r04_pmc_summary.txt" "k_panel128_h<true, 0, 1, 2>=2" "k_panel128_h<false, 0, 1, 2>=1" "k_panel128_h<false, 1, 2, 2>=1" "k_panel128_h<true, 2, 3, 2>=1" "k_res64_lds<true, 0>=2" "k_res64_lds<true, 4>=1" "k_res64_dual=1" "k_fused_narrow_lds<2>=2" "k_linear_h<4, 1, 0, false>=1" "k_update=1"
rm -rf $OUT/pmc/p1 $OUT/pmc/p2 $OUT/pmc/p3 $OUT/pmc/p4 $OUT/pmc/p5 $OUT/pmc/p6
