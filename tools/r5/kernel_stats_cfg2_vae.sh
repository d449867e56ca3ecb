export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/prof_r05b; mkdir -p $O; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_cfg2 -o k -- python $R/tools/bench_parts.py cfg3 default-only > $O/cfg2.json 2> $O/cfg2.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_vae -o k -- python $R/tools/bench_parts.py vae > $O/vae.json 2> $O/vae.err
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head
