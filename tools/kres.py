#!/usr/bin/env python3
"""Prints VGPR / AGPR / spill / scratch / occupancy of every tconv instantiation (hipcc remarks)."""
import re, subprocess, sys, os
C = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "latent_diffusion_planning_amd", "csrc")
for f in sys.argv[1:] or ["tconv_k5", "tconv_k5r", "tconv_misc"]:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f"{C}/{f}.hip", "-o", "/tmp/_k.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            n = m.group(1)
            mm = re.search(r"tconv_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", n)
            cur = "tconv<%s,%s,%s,%s,%s,%s>" % mm.groups() if mm else n[:50]
            vals = {}
        for key in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur:
                vals[key] = m.group(1)
        if "LDS Size" in line and cur:
            print(f"{cur:26s} vgpr={vals.get('VGPRs'):>4s} agpr={vals.get('AGPRs','-'):>3s} spill={vals.get('VGPRs Spill'):>3s} scratch={vals.get('ScratchSize [bytes/lane]'):>4s} occ={vals.get('Occupancy [waves/SIMD]')}")
            cur = None
