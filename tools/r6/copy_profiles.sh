#!/bin/bash
# gpurun_out/prof_r06 -> profiles/r06_* (the judged, committed copies)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd); P=$R/gpurun_out/prof_r06; D=$R/profiles
cp $P/bench.json $D/r06_bench_b256_ddim100.json
cp $P/ks_bench/k_kernel_stats.csv $D/r06_kernel_stats_b256_ddim100.csv
cp $P/pmc_summary.json $D/r06_pmc_b256_ddim100.json
cp $P/layer_times_b256.txt $D/r06_layer_times_b256.txt
cp $P/bench_config2.json $D/r06_bench_config2_rm_square_t16_b1024.json
cp $P/bench_config3.json $D/r06_bench_config3_aloha_b512.json
cp $P/bench_config4.json $D/r06_bench_config4_rm_can_b1024_ddim50.json
cp $P/ks_config2/k_kernel_stats.csv $D/r06_kernel_stats_config2_t16_b1024.csv
cp $P/ks_config3/k_kernel_stats.csv $D/r06_kernel_stats_config3_aloha_b512.csv
cp $P/ks_config4/k_kernel_stats.csv $D/r06_kernel_stats_config4_b1024_ddim50.csv
cp $P/bench_train.json $D/r06_update_bench_b256.json
cp $P/ks_train/k_kernel_stats.csv $D/r06_kernel_stats_update_b256.csv
cp $P/other_configs.json $D/r06_other_configs.json
cp $P/parity_margins.json $D/r06_parity_margins.json
grep -v amdgpu.ids $P/stress_exchange.txt > $D/r06_exchange_soak.txt
ls -la $D | grep r06
