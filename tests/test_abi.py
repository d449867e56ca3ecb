"""The C-ABI shared library loads and exports exactly what include/ldp_hip.h declares (CPU)."""
import ctypes
import os
import subprocess
import sys

import pytest

from latent_diffusion_planning_amd import _lib


def _built():
    return os.path.exists(_lib.LIB_PATH)


def test_header_and_binding_list_the_same_symbols():
    assert _lib.header_symbols() == sorted(_lib.SIGNATURES)


@pytest.mark.skipif(not _built(), reason="libldp_hip.so not built (run __graft_entry__.build())")
def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _lib.header_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.ldp_version()


@pytest.mark.skipif(not _built(), reason="libldp_hip.so not built")
def test_the_loaded_library_was_built_from_this_tree():
    """ldp_version() carries a hash of csrc/*.hip, csrc/*.hpp and include/ldp_hip.h (csrc/Makefile); recomputed from
    the tree it must match -- so a test log on the GPU box says which source state was mapped (the .so is not in git)."""
    assert _lib.built_source_hash() == _lib.source_hash(), (
        f"libldp_hip.so reports {_lib.load().ldp_version()!r} but the tree hashes to {_lib.source_hash()}: rebuild "
        "(python -c 'import __graft_entry__ as g; g.build()')")


@pytest.mark.skipif(not _built(), reason="libldp_hip.so not built")
def test_bad_arguments_return_codes_without_a_gpu():
    lib = _lib.load()
    assert lib.ldp_create(None, None) == -1                       # LDP_EINVAL, no HIP call made
    assert b"null" in lib.ldp_last_error()
    assert lib.ldp_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LDPHipUnavailable, match="only compute path"):
        _lib.load()


@pytest.mark.skipif(not _built(), reason="libldp_hip.so not built")
def test_the_dynamic_symbol_table_is_exactly_the_header():
    """-fvisibility=hidden + csrc/exports.map -- `nm -D --defined-only` lists the C ABI and nothing else (no C++ internals, no
    kernel stubs)."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(_lib.SIGNATURES), sorted(set(exported) ^ set(_lib.SIGNATURES))


@pytest.mark.skipif(not _built() or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the built library and llvm-objdump")
def test_no_packed_fp32_operand_selection_in_the_shipped_code_objects():
    """Round 6: on gfx950 a packed-fp32 instruction whose LOW lane reads the HIGH dword of a source (`op_sel` with a 1) can read 0.0 in
    lanes 48..63 next to another wave's v_mfma_f32_16x16x32_f16 (tools/r6/pk_f32_repro.hip reproduces it stand-alone; timing-dependent, so no soak
    clears a binary that contains the form).  The library is compiled without packed-fp32 instructions; this audits what was actually built."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "r6"))
    import pk_audit
    bad = pk_audit.offenders(_lib.LIB_PATH)
    assert not bad, {k: v[:2] for k, v in bad.items()}


def test_package_version_is_the_library_version():
    import latent_diffusion_planning_amd as pkg
    assert pkg.__version__ == pkg.library_version_from_source()
    if _built():
        assert ("ldp_hip " + pkg.__version__).encode() in _lib.load().ldp_version()
