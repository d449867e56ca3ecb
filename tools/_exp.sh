R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/exp_$1
for d in 16 24; do LDP_DBG=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/exp_$1/dbg$d -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/exp_$1/dbg$d.log 2>&1; done
