#!/bin/bash
# Regenerates the round's evidence under gpurun_out/prof_$1 on the GPU box (copy what is to be judged into
# profiles/ afterwards).  Usage: tools/make_profiles.sh r02
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# 1. the driver's bench line
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. per-kernel time of the same command (short run)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_bench -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ks_bench.log 2>&1
# 3. PMC passes of the same command (separate passes) + summary
bash $R/tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
# 4. the other configurations / pieces
python $R/tools/bench_parts.py idm vae cfg3 cfg4 cfg5 agent > $OUT/other_configs.json 2> $OUT/other_configs.err
# 5. kernel stats of the fused IDM loop, the joint T=16 graph, the VAE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_idm -o k -- python $R/tools/bench_parts.py idm256 > $OUT/ks_idm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg3 -o k -- python $R/tools/bench_parts.py cfg3 default-only > $OUT/ks_cfg3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_vae -o k -- python $R/tools/bench_parts.py vae > $OUT/ks_vae.log 2>&1
# 6. parity margins against every golden
python $R/tools/parity_margin.py $OUT/parity_margins.json > $OUT/parity_margins.log 2>&1
# 7. ablation of the conv launch (untraced wall clock per launch)
ABL=$R/latent_diffusion_planning_amd/libldp_hip_abl.so      # the dbg switches exist in the ablation build only (make ablate)
[ -f $ABL ] || make -C $R/latent_diffusion_planning_amd/csrc -j16 ablate > $OUT/ablate_build.log 2>&1
for d in 0 64 24 8 16 32 128; do
  python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lib $ABL --opt dbg=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($d, d['roofline']['avg_launch_us'])"
done > $OUT/ablation_untraced.txt 2>&1
# 8. round 3: per-wave timeline of one evaluation (instrumented library), per-layer launch times, graph capture
#    cost, and the soak tools' counts (stress of the in-launch exchanges; two engine processes sharing the GPU)
make -C $R/latent_diffusion_planning_amd/csrc timeline -j16 > $OUT/timeline_build.log 2>&1
python $R/tools/timeline.py --json $OUT/timeline_b256.json > $OUT/timeline_b256.txt 2>&1
f=$(find $OUT/ks_bench -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/layer_times.py $f 30 256 > $OUT/layer_times_b256.txt 2>&1
python $R/tools/graph_cost.py > $OUT/graph_cost.json 2> $OUT/graph_cost.err
python $R/tools/stress_exchange.py 100 > $OUT/stress_exchange.txt 2>&1
( python $R/tools/shared_gpu_check.py 16 150 > $OUT/shared_gpu_a.txt 2>&1 & python $R/tools/shared_gpu_check.py 64 150 > $OUT/shared_gpu_b.txt 2>&1; wait )
# 9. round 4: split-operand StableVAE A/B (same process), small-batch HBM roofline, configs[0] CPU vs GPU, the floor model
#    of an evaluation (kernel traces of the ablation build), the split conv's PMC pass + ablation, the VAE margins
python $R/tools/bench_parts.py vae_ab > $OUT/split_vae_ab.json 2> $OUT/split_vae_ab.err
python $R/tools/bench_parts.py small > $OUT/small_batch.json 2> $OUT/small_batch.err
python $R/bench.py --configs0 > $OUT/configs0.json 2> $OUT/configs0.err
mkdir -p $OUT/floor
for d in 0 16 24 64; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/floor/dbg$d -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lib $ABL --opt dbg=$d > $OUT/floor/dbg$d.log 2>&1
done
python $R/tools/r4/floor_model.py $OUT/floor 256 > $OUT/floor_model.txt 2>&1
bash $R/tools/r4/sconv_ablate.sh > $OUT/sconv_ablate_and_pmc.txt 2>&1
mkdir -p $R/gpurun_out/r4
( cd $R && timeout 600 python -m pytest tests/test_hip_vae.py -q -m gpu -k margins > $OUT/vae_margins_test.txt 2>&1 ); cp $R/gpurun_out/r4/vae_margins.json $OUT/split_vae_margins.json 2>/dev/null
$R/tools/bin/split_bf16_probe > $OUT/split_probe.txt 2>&1
# 10. round 4: the planner on split operands -- same-box A/B by horizon and batch, per-layer times, matrix-pipe busy fraction,
#     ablations of the split main loop (ablation build: the switches perturb the schedule, read the differences), margins
{ for tb in "8 1024 ddim 50" "16 1024 ddim 50" "8 512 ddim 50" "16 512 ddim 50" "8 2048 ddim 50" "16 320 ddim 50"; do python $R/tools/r4/psplit.py $tb; done; } 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/split_planner_ab.txt
{ bash $R/tools/r4/ps_stats.sh 1024; bash $R/tools/r4/ps_stats.sh 512; } > $OUT/split_planner_layers.txt 2>&1
{ bash $R/tools/r4/ps_pmc.sh 16 1024; bash $R/tools/r4/ps_pmc.sh 8 1024; } > $OUT/split_planner_pmc.txt 2>&1
{ DBGS="0 512 1024 1536 8 16" bash $R/tools/r4/ps_ablate.sh 16 1024; DBGS="0 512 1024 1536 8 16" bash $R/tools/r4/ps_ablate.sh 8 1024; } > $OUT/split_planner_ablation.txt 2>&1
# the nine-product (exact-product) and three-product forms of the 16-row tiles against the six-product default and exact fp32, at 256 plans
# (option planner_split = 3) and at 1024 plans (`make -C csrc products` builds the A/B libraries; skipped when they are absent)
for l in libldp_hip.so libldp_hip_p9.so libldp_hip_p3.so; do [ -f $R/latent_diffusion_planning_amd/$l ] && python $R/tools/r4/nine.py $R/latent_diffusion_planning_amd/$l 2>&1 | grep -v "Warning\|amdgpu.ids"; done > $OUT/split_planner_products_ab.txt
rm -f $R/gpurun_out/r4/planner_split_margins.json
( cd $R && timeout 600 python -m pytest tests/test_hip_planner.py -q -m gpu -k split_operands > $OUT/split_planner_margins_test.txt 2>&1 ); cp $R/gpurun_out/r4/planner_split_margins.json $OUT/split_planner_margins.json 2>/dev/null
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT
