set -u
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_vae.py -x -q -m gpu -k "split or conv3x3 or encode or decode" > gpurun_out/r4/t_vae.txt 2>&1
tail -15 gpurun_out/r4/t_vae.txt
timeout 600 python tools/bench_parts.py vae_ab > gpurun_out/r4/vae_ab.json 2> gpurun_out/r4/vae_ab.err
cat gpurun_out/r4/vae_ab.json | head -60
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/ks_vae_ab -o k -- python $GRAFT_REPO_ROOT/tools/bench_parts.py vae_ab > $GRAFT_REPO_ROOT/gpurun_out/r4/ks_vae_ab.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r4/ks_vae_ab -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 $f
find gpurun_out/r4 -name "*_kernel_trace.csv" -size +20M -delete
