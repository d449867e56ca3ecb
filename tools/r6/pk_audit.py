#!/usr/bin/env python3
"""Round 6: audit of the shipped code objects for the packed-fp32 operand selection that tools/r6/pk_f32_repro.hip shows returning a wrong source
(0.0) in lanes 48..63 next to another wave's v_mfma_f32_16x16x32_f16: a VOP3P *_f32 instruction whose LOW result lane reads the HIGH dword of a
64-bit source (op_sel with a 1 in it).  Extracts every gfx950 code object from the .so's clang offload bundles, disassembles it with llvm-objdump
and lists `v_pk_(add|mul|fma)_f32` with such an op_sel per kernel.   python tools/r6/pk_audit.py [path/to/libldp_hip.so]  -> exit 1 if any."""
import os, re, struct, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + 24


def offenders(lib):
    """-> {kernel: [instruction text]}"""
    out = {}
    for k, (triple, co) in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            name = f.name
        try:
            txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(name)
        kernel = "?"
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
                continue
            if PK.search(line):
                s = OPSEL.search(line)
                if s and "1" in s.group(1):
                    out.setdefault(kernel, []).append(line.split("//")[0].strip())
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "latent_diffusion_planning_amd", "libldp_hip.so")
    bad = offenders(lib)
    n = sum(len(v) for v in bad.values())
    print(f"{os.path.basename(lib)}: {n} packed-fp32 instructions whose LOW lane reads a HIGH dword (op_sel) in {len(bad)} kernels")
    for k, v in sorted(bad.items()):
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"  {len(v):3d}  {dem[:150]}")
        for ins in v[:3]:
            print(f"         {ins}")
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
