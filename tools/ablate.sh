#!/bin/bash
# Kernel-time ablation of the conv kernels (results are wrong by construction; only timings matter):
# option dbg: bit 8 = no main loop, 16 = no epilogue, 32 = no GroupNorm statistics exchange, 64 = empty kernel.
# The switches live in the ablation build only: `make -C latent_diffusion_planning_amd/csrc ablate` -> libldp_hip_abl.so.
R=$(cd "$(dirname "$0")/.." && pwd)
ABL=$R/latent_diffusion_planning_amd/libldp_hip_abl.so
[ -f $ABL ] || make -C $R/latent_diffusion_planning_amd/csrc -j16 ablate > /dev/null
OUT=$1; shift; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for d in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dbg$d -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --lib $ABL --opt dbg=$d > $OUT/dbg$d.log 2>&1
done
python3 - "$OUT" "$@" <<'PY'
import csv, re, sys
out, ds = sys.argv[1], sys.argv[2:]
tab = {}
for d in ds:
    for r in csv.DictReader(open(f"{out}/dbg{d}/a_kernel_stats.csv")):
        m = re.search(r"tconv_kernel<(.*?)>", r["Name"])
        if m:
            tab.setdefault(m.group(1).replace(" ", ""), {})[d] = float(r["AverageNs"]) / 1e3
print("kernel".ljust(22), *[("dbg" + d).rjust(7) for d in ds])
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get(ds[0], 0)):
    print(k.ljust(22), *[f"{v.get(d, 0):7.1f}" for d in ds])
PY
