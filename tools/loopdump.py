#!/usr/bin/env python3
"""Compressed view of the main-loop instruction stream of one tconv instantiation (M = mfma run)."""
import re, subprocess, sys, os
C = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "latent_diffusion_planning_amd", "csrc")
unit, key = sys.argv[1], sys.argv[2]      # e.g. tconv_k5 Li0ELi2ELi8ELi1ELi4ELb0
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", f"-I{C}", f"{C}/{unit}.hip", "-o", "/tmp/_l.s"], capture_output=True)
s = open("/tmp/_l.s").read()
parts = re.split(r'\n\s*\.type\s+(_ZN3ldp12tconv_kernel\w+),@function', s)
for i in range(1, len(parts), 2):
    if key not in parts[i]:
        continue
    body = parts[i + 1].split('.end_amdhsa_kernel')[0]
    lines = [l.strip() for l in body.split('\n')]
    start = [k for k, l in enumerate(lines) if l.startswith('ds_read_b128')][0]
    out = []
    for l in lines[max(0, start - 40):start + int(sys.argv[3]) if len(sys.argv) > 3 else start + 400]:
        if re.match(r'(v_mfma|global_load|ds_|s_waitcnt|s_barrier|s_cbranch|\.LBB|v_mov_b32|scratch_)', l):
            out.append(l.split(';')[0][:60])
    res, run = [], 0
    for l in out:
        if l.startswith('v_mfma'):
            run += 1
            continue
        if run:
            res.append(f"   [{run} x v_mfma]")
            run = 0
        res.append(l)
    if run:
        res.append(f"   [{run} x v_mfma]")
    print('\n'.join(res))
