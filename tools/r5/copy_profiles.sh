#!/bin/bash
# gpurun_out/prof_r05 -> profiles/r05_* (the judged, committed copies)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd); P=$R/gpurun_out/prof_r05; D=$R/profiles
cp $P/bench.json $D/r05_bench_b256_ddim100.json
cp $P/ks_bench/k_kernel_stats.csv $D/r05_kernel_stats_b256_ddim100.csv
cp $P/pmc_summary.json $D/r05_pmc_b256_ddim100.json
cp $P/bench_config3.json $D/r05_bench_config3_aloha_b512.json
cp $P/bench_config4.json $D/r05_bench_config4_rm_can_b1024_ddim50.json
cp $P/ks_config3/k_kernel_stats.csv $D/r05_kernel_stats_config3_aloha_b512.csv
cp $P/ks_config4/k_kernel_stats.csv $D/r05_kernel_stats_config4_b1024_ddim50.csv
cp $P/other_configs.json $D/r05_other_configs.json
cp $P/parity_margins.json $D/r05_parity_margins.json
cp $P/layer_times_b256.txt $D/r05_layer_times_b256.txt
grep -v amdgpu.ids $P/stress_exchange.txt > $D/r05_exchange_soak.txt
ls -la $D | grep r05
