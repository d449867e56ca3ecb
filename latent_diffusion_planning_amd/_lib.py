"""ctypes binding of libldp_hip.so (the C ABI in include/ldp_hip.h).

There is no fallback of any kind: if the shared library is missing or does not load, importing
the compute path raises `LDPHipUnavailable` with the build command.  torch is imported first so
that the library binds to the same libamdhip64 instance PyTorch-ROCm already loaded (device
pointers and streams are then shared).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldp_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ldp_hip.h")

LDP_MAX_LEVELS = 4
SAMPLER_DDPM, SAMPLER_DDIM = 0, 1
MOD_PLANNER, MOD_IDM, MOD_VAE = 1, 2, 4


class LDPHipUnavailable(RuntimeError):
    pass


class LDPHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libldp_hip error {code}: {msg}")
        self.code = code


class LDPHipFault(LDPHipError):
    """LDP_EFAULT: results since the last poll are invalid -- a split work-group timed out on its peer, or an operand
    left the range of the two-fp16-plane convolutions (include/ldp_hip.h, fault protocol)."""


LDP_EFAULT = -6


class LdpConfig(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("action_dim", C.c_int32), ("global_cond_dim", C.c_int32),
        ("pred_horizon", C.c_int32), ("action_horizon", C.c_int32), ("n_levels", C.c_int32),
        ("down_dims", C.c_int32 * LDP_MAX_LEVELS), ("kernel_size", C.c_int32),
        ("n_groups", C.c_int32), ("step_embed_dim", C.c_int32),
        ("planner_train_steps", C.c_int32), ("idm_train_steps", C.c_int32),
        ("idm_hidden", C.c_int32), ("idm_blocks", C.c_int32), ("idm_time_dim", C.c_int32),
        ("image_size", C.c_int32), ("vae_latent_channels", C.c_int32), ("device", C.c_int32),
    ]


_FP = C.c_void_p       # device float*/int* passed as integers from tensor.data_ptr()
_H = C.c_void_p        # ldp_handle*

# name -> (restype, argtypes); must list every symbol include/ldp_hip.h declares
SIGNATURES: Dict[str, tuple] = {
    "ldp_last_error": (C.c_char_p, []),
    "ldp_version": (C.c_char_p, []),
    "ldp_create": (C.c_int, [C.POINTER(LdpConfig), C.POINTER(_H)]),
    "ldp_destroy": (C.c_int, [_H]),
    "ldp_set_weight": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "ldp_finalize": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "ldp_unet_forward": (C.c_int, [_H, _FP, _FP, C.c_int32, _FP, _FP, C.c_int32, C.c_void_p]),
    "ldp_plan_sample": (C.c_int, [_H, _FP, _FP, _FP, C.c_uint64, C.c_int64, C.c_int32, C.c_int32,
                                  _FP, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_idm_forward": (C.c_int, [_H, _FP, _FP, _FP, C.c_int32, _FP, C.c_int32, C.c_void_p]),
    "ldp_idm_sample": (C.c_int, [_H, _FP, _FP, _FP, C.c_uint64, C.c_int64, C.c_int32, C.c_int32,
                                 _FP, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_vae_encode": (C.c_int, [_H, _FP, _FP, C.c_int32, C.c_void_p]),
    "ldp_vae_decode": (C.c_int, [_H, _FP, _FP, C.c_int32, C.c_void_p]),
    "ldp_conv2d_3x3_f32": (C.c_int, [_FP, C.c_void_p, C.c_void_p, _FP, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_conv2d_3x3_bf16x3": (C.c_int, [_FP, C.c_void_p, C.c_void_p, _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_normalize_bounds": (C.c_int, [_FP, _FP, C.c_int64, _FP, _FP, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_mean_sq_diff": (C.c_int, [_FP, _FP, C.c_int64, _FP, C.c_void_p]),
    "ldp_conv1d_gn_mish_film_f32": (C.c_int, [_FP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _FP,
                                              _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_downsample1d_f32": (C.c_int, [_FP, C.c_void_p, C.c_void_p, _FP, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]),
    "ldp_upsample1d_f32": (C.c_int, [_FP, C.c_void_p, C.c_void_p, _FP, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "ldp_agent_sample": (C.c_int, [_H, _FP, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, C.c_uint64, C.c_int64,
                                   C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p]),
    "ldp_check_fault": (C.c_int, [_H, C.c_void_p]),
    "ldp_poll_fault": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "ldp_set_option": (C.c_int, [_H, C.c_char_p, C.c_int64]),
    "ldp_get_option": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int64)]),
    "ldp_philox_raw": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _FP, C.c_int64, C.c_void_p]),
    "ldp_philox_normal": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _FP, C.c_int64, C.c_void_p]),
    "ldp_launch_count": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64)]),
    "ldp_range_fallbacks": (C.c_int64, []),
    "ldp_add_noise": (C.c_int, [_FP, _FP, _FP, C.c_int32, _FP, C.c_int64, C.c_int32, C.c_void_p]),
    "ldp_reduce_stats": (C.c_int, [_FP, C.c_int64, _FP, C.c_void_p]),
    "ldp_train_init": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "ldp_train_planner_grad": (C.c_int, [_H, _FP, _FP, _FP, _FP, C.c_float, _FP, C.c_int32, C.c_void_p]),
    "ldp_train_idm_grad": (C.c_int, [_H, _FP, _FP, _FP, _FP, C.c_float, _FP, C.c_int32, C.c_void_p]),
    "ldp_train_grad_norm": (C.c_int, [_H, C.c_int32, _FP, C.c_void_p]),
    "ldp_train_apply": (C.c_int, [_H, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "ldp_train_step_count": (C.c_int, [_H, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]),
    "ldp_train_read": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ldp_train_write": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ldp_train_arena": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "ldp_train_publish": (C.c_int, [_H, C.c_int32, C.c_void_p]),
}


def header_symbols(path: str = HEADER_PATH) -> List[str]:
    """Every function name declared in include/ldp_hip.h."""
    with open(path) as f:
        txt = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ldp_[a-z0-9_]+)\s*\(", txt)))


def source_hash() -> str:
    """What csrc/Makefile bakes into ldp_version(): sha256 over csrc/*.hip, csrc/*.hpp (byte-sorted names) and
    include/ldp_hip.h, first 16 hex digits -- recomputed from the tree."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") or f.endswith(".hpp"))
    h = hashlib.sha256()
    for f in names:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    with open(HEADER_PATH, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def built_source_hash() -> str:
    """The source hash the loaded library reports (ldp_version(): '... src:<hash> [flavour]')."""
    m = re.search(rb"src:([0-9a-f]{16})", load().ldp_version())
    return m.group(1).decode() if m else ""


_lib = None


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises LDPHipUnavailable loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LDPHipUnavailable(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of this package. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C latent_diffusion_planning_amd/csrc -j`).")
    try:
        import torch  # noqa: F401  (loads libamdhip64 / libhsa-runtime64 first)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise LDPHipUnavailable(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LDPHipUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load().ldp_last_error()
        cls = LDPHipFault if code == LDP_EFAULT else LDPHipError
        raise cls(code, msg.decode() if msg else "?")
