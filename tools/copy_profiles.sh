#!/bin/bash
# Copies the summaries tools/make_profiles.sh left under gpurun_out/prof_$1 into profiles/ (the judged, committed copies).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:-r04}; P=$R/gpurun_out/prof_$TAG; D=$R/profiles
cp $P/bench.json $D/${TAG}_bench_b256_ddim100.json
cp $P/ks_bench/k_kernel_stats.csv $D/${TAG}_kernel_stats_b256_ddim100.csv
cp $P/pmc_summary.json $D/${TAG}_pmc_b256_ddim100.json
cp $P/ks_idm/k_kernel_stats.csv $D/${TAG}_kernel_stats_idm_loop_b256.csv
cp $P/ks_cfg3/k_kernel_stats.csv $D/${TAG}_kernel_stats_cfg3_t16_b1024_joint_graph.csv
cp $P/ks_vae/k_kernel_stats.csv $D/${TAG}_kernel_stats_vae_enc_dec.csv
cp $P/other_configs.json $D/${TAG}_other_configs.json
cp $P/parity_margins.json $D/${TAG}_parity_margins.json
# (the per-wave timeline is not copied since round 4: its cross-XCD clock alignment gives negative gaps on this image; profiles/r04_floor_model.txt is the per-launch breakdown)
cp $P/layer_times_b256.txt $D/${TAG}_layer_times_b256.txt
cp $P/graph_cost.json $D/${TAG}_graph_capture_cost.json
cp $P/ablation_untraced.txt $D/${TAG}_conv_launch_ablation.txt
# the soak file also holds hand-run long soaks (profiles/README.md): only written when absent
[ -e $D/${TAG}_exchange_soak.txt ] || { grep -v amdgpu.ids $P/stress_exchange.txt; echo "--- two engine processes sharing the GPU (tools/shared_gpu_check.py 16 150 & ... 64 150):"
  grep -v amdgpu.ids $P/shared_gpu_a.txt; grep -v amdgpu.ids $P/shared_gpu_b.txt; } > $D/${TAG}_exchange_soak.txt
# round 4
for f in split_vae_ab.json small_batch.json split_vae_margins.json floor_model.txt split_probe.txt split_planner_ab.txt split_planner_layers.txt split_planner_pmc.txt split_planner_ablation.txt split_planner_margins.json split_planner_products_ab.txt; do [ -s $P/$f ] && cp $P/$f $D/${TAG}_$f; done
[ -s $P/configs0.json ] && cp $P/configs0.json $D/${TAG}_configs0_cpu_vs_gpu.json
[ -s $P/sconv_ablate_and_pmc.txt ] && grep -v amdgpu.ids $P/sconv_ablate_and_pmc.txt > $D/${TAG}_split_conv_ablation_and_pmc.txt
bash $R/tools/resource_usage.sh > $D/${TAG}_kernel_resource_usage.txt 2>/dev/null || true
