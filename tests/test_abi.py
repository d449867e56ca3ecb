"""The C-ABI shared library loads and exports exactly what include/ldp_hip.h declares (CPU)."""
import ctypes
import os

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
