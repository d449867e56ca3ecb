#!/bin/bash
# Per-instantiation register / LDS / occupancy table of the tconv kernels (VERDICT r2 #1d): compiles the four
# instantiation units with -Rpass-analysis=kernel-resource-usage and condenses the remarks.
# usage: tools/resource_usage.sh > profiles/r05_kernel_resource_usage.txt
R=$(cd "$(dirname "$0")/.." && pwd); cd $R/latent_diffusion_planning_amd/csrc
echo "# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage ; tconv_kernel<MODE,TO,NWN,KS,CPI,RES_OUT,MB,KWS,SPLIT>"
echo "# MODE: 0 k5, 1 stride-2, 2 transposed, 3 1x1, 4 3x3 (2-D), 5 3x3 stride 2 (2-D)"
printf "%-34s %6s %6s %6s %8s %8s %10s\n" instantiation VGPRs AGPRs SGPRs spill_B LDS_B "waves/SIMD"
for f in tconv_k5 tconv_k5r tconv_misc tconv_2d tconv_split tconv_split16a tconv_split16b tconv_split16c tconv_split2 tconv_split3 tconv_split3b idm sconv; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage $f.hip -o /dev/null 2>&1 | \
  python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\d+)", line)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    m = re.search(r"(tconv_kernel|idm_block_kernel|idm_block_h16_kernel|sconv3_kernel|planes_kernel)<(.*?)>", name)
    if not m: continue
    tag = m.group(1).replace("_kernel", "") + "<" + m.group(2).replace(" ", "").replace("false", "0").replace("true", "1") + ">"
    print("%-34s %6d %6d %6d %8d %8d %10d" % (tag, v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize [bytes/lane]", 0), v.get("LDS Size [bytes/block]", 0), v.get("Occupancy [waves/SIMD]", 0)))
'
done
echo "# LDS is dynamic (TConvCfg::LDS_BYTES, set by hipFuncSetAttribute): 2 x staged tile or the epilogue tile, 16-128 KiB; the static figure above is 0 for tconv"
