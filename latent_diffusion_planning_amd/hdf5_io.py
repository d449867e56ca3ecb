"""Minimal HDF5 writer/reader on the HDF5 C library through ctypes -- enough for the `latent.hdf5` container the
reference's data layer reads (process_sdvae_data.py:57-59,110-119 writes it with h5py; consumer
data/robomimic_latent_data.py:94-96: `latent_file['data'][demo]['latent'][key][:]`):

    /data                      group, attributes `total` (int64), `min_z`, `max_z` (float32)
    /data/<ep>/latent/<key>    float32 dataset (T+1, h, w, c), contiguous layout

h5py is not installed for this interpreter, but the C library it wraps ships with the image (libhdf5.so of the conda
tree); any libhdf5 >= 1.10 found by the loader works.  No fallback container: `HDF5Unavailable` is raised when no
library can be loaded (preencode.save_latents(..., fmt="npz") remains for such machines).  Host-side file I/O only --
nothing here is on the sampling path.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Iterable, Optional

import numpy as np

hid_t = C.c_int64          # HDF5 >= 1.10; 1.8's 32-bit hid_t is refused by load() (the signatures below would be wrong)
herr_t = C.c_int
hsize_t = C.c_uint64

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SCALAR = 0


class _GInfo(C.Structure):          # H5G_info_t
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


class HDF5Unavailable(RuntimeError):
    pass


class HDF5Error(RuntimeError):
    pass


_lib = None


def _candidates() -> Iterable[str]:
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "fortran" not in p:
                yield p


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    errs = []
    for path in _candidates():
        try:
            lib = C.CDLL(path)
            lib.H5open.restype = herr_t
            if lib.H5open() < 0:
                raise OSError("H5open failed")
            ver = (C.c_uint * 3)()
            lib.H5get_libversion(C.byref(ver, 0), C.byref(ver, 4), C.byref(ver, 8))
            if (ver[0], ver[1]) < (1, 10):
                raise OSError(f"HDF5 {ver[0]}.{ver[1]}.{ver[2]} has a 32-bit hid_t; 1.10 or newer is needed")
        except OSError as e:
            errs.append(f"{path}: {e}")
            continue
        sig = {
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gclose": (herr_t, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aclose": (herr_t, [hid_t]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]), "H5Pclose": (herr_t, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Gget_info": (herr_t, [hid_t, C.POINTER(_GInfo)]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
            "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib.H5Eset_auto2.argtypes = [hid_t, C.c_void_p, C.c_void_p]
            lib.H5Eset_auto2(0, None, None)             # errors surface as HDF5Error, not as stacks on stderr
            lib._t_f32 = hid_t.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
            lib._t_i64 = hid_t.in_dll(lib, "H5T_NATIVE_LLONG_g").value
            lib._t_u8 = hid_t.in_dll(lib, "H5T_NATIVE_UCHAR_g").value
            lib._p_lcpl = hid_t.in_dll(lib, "H5P_CLS_LINK_CREATE_ID_g").value
        except (AttributeError, ValueError) as e:
            errs.append(f"{path}: {e}")
            continue
        _lib = lib
        return lib
    raise HDF5Unavailable("no usable HDF5 C library (libhdf5.so) found: " + ("; ".join(errs) or "none on the search path") +
                          ".  latent.hdf5 cannot be written here; use preencode.save_latents(..., fmt='npz')")


def _ok(v, what):
    if v < 0:
        raise HDF5Error(f"HDF5: {what} failed ({v})")
    return v


class File:
    """Context manager over one HDF5 file: float32 datasets at slash-separated paths (intermediate groups are
    created), scalar attributes on groups."""

    def __init__(self, path: str, mode: str = "r"):
        self.lib = load()
        p = os.fsencode(path)
        if mode == "w":
            self.fid = _ok(self.lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
        elif mode == "r":
            self.fid = _ok(self.lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
        else:
            raise ValueError("mode must be 'r' or 'w'")
        try:
            self._lcpl = _ok(self.lib.H5Pcreate(self.lib._p_lcpl), "link creation property list")
            _ok(self.lib.H5Pset_create_intermediate_group(self._lcpl, 1), "create_intermediate_group")
        except HDF5Error:
            self.lib.H5Fclose(self.fid)
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.fid is not None:
            self.lib.H5Pclose(self._lcpl)
            _ok(self.lib.H5Fclose(self.fid), "close")
            self.fid = None

    # -- writing ---------------------------------------------------------------------------------------
    def require_group(self, name: str) -> None:
        n = name.strip("/").encode()
        if self.lib.H5Lexists(self.fid, n, H5P_DEFAULT) > 0:
            return
        g = _ok(self.lib.H5Gcreate2(self.fid, n, self._lcpl, H5P_DEFAULT, H5P_DEFAULT), f"create group {name}")
        self.lib.H5Gclose(g)

    def write_dataset(self, name: str, array, dtype=np.float32) -> None:
        """float32 (latents) or uint8 (camera frames, the robomimic input files) dataset, contiguous layout."""
        if np.dtype(dtype) not in (np.dtype(np.float32), np.dtype(np.uint8)):
            raise ValueError("hdf5_io writes float32 or uint8 datasets")
        a = np.ascontiguousarray(np.asarray(array), dtype=dtype)
        ft = self.lib._t_f32 if a.dtype == np.float32 else self.lib._t_u8
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        space = _ok(self.lib.H5Screate_simple(a.ndim, dims, None) if a.ndim else self.lib.H5Screate(H5S_SCALAR), "dataspace")
        d = _ok(self.lib.H5Dcreate2(self.fid, name.strip("/").encode(), ft, space, self._lcpl, H5P_DEFAULT, H5P_DEFAULT),
                f"create dataset {name}")
        try:
            if a.size:
                _ok(self.lib.H5Dwrite(d, ft, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), f"write {name}")
        finally:
            self.lib.H5Dclose(d)
            self.lib.H5Sclose(space)

    def write_attr(self, group: str, name: str, value) -> None:
        """Scalar attribute: Python int -> int64, anything else -> float32 (what h5py stores for a jnp.float32 scalar)."""
        is_int = isinstance(value, (int, np.integer)) and not isinstance(value, bool)
        buf = (C.c_int64 if is_int else C.c_float)(int(value) if is_int else float(value))
        t = self.lib._t_i64 if is_int else self.lib._t_f32
        g = _ok(self.lib.H5Gopen2(self.fid, group.strip("/").encode(), H5P_DEFAULT), f"open group {group}")
        space = _ok(self.lib.H5Screate(H5S_SCALAR), "scalar dataspace")
        at = _ok(self.lib.H5Acreate2(g, name.encode(), t, space, H5P_DEFAULT, H5P_DEFAULT), f"create attribute {name}")
        try:
            _ok(self.lib.H5Awrite(at, t, C.byref(buf)), f"write attribute {name}")
        finally:
            self.lib.H5Aclose(at)
            self.lib.H5Sclose(space)
            self.lib.H5Gclose(g)

    # -- reading (round-trip checks) ---------------------------------------------------------------------
    def keys(self, group: str):
        """Link names of a group (h5py's `list(f[group].keys())`: increasing name order)."""
        g = _ok(self.lib.H5Gopen2(self.fid, group.strip("/").encode(), H5P_DEFAULT), f"open group {group}")
        try:
            info = _GInfo()
            _ok(self.lib.H5Gget_info(g, C.byref(info)), "group info")
            out = []
            for i in range(int(info.nlinks)):          # H5_INDEX_NAME = 0, H5_ITER_INC = 0
                n = _ok(self.lib.H5Lget_name_by_idx(g, b".", 0, 0, i, None, 0, H5P_DEFAULT), "link name")
                buf = C.create_string_buffer(n + 1)
                _ok(self.lib.H5Lget_name_by_idx(g, b".", 0, 0, i, buf, n + 1, H5P_DEFAULT), "link name")
                out.append(buf.value.decode())
            return out
        finally:
            self.lib.H5Gclose(g)

    def exists(self, name: str) -> bool:
        parts, cur = name.strip("/").split("/"), ""
        for p in parts:                                 # H5Lexists needs every intermediate link to exist
            cur = f"{cur}/{p}" if cur else p
            if self.lib.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def shape(self, name: str):
        d = _ok(self.lib.H5Dopen2(self.fid, name.strip("/").encode(), H5P_DEFAULT), f"open dataset {name}")
        space = _ok(self.lib.H5Dget_space(d), "dataspace")
        try:
            nd = _ok(self.lib.H5Sget_simple_extent_ndims(space), "rank")
            dims = (hsize_t * max(nd, 1))()
            if nd:
                _ok(self.lib.H5Sget_simple_extent_dims(space, dims, None), "dims")
            return tuple(int(x) for x in dims[:nd])
        finally:
            self.lib.H5Sclose(space)
            self.lib.H5Dclose(d)

    def read_dataset(self, name: str, start: int = 0, count: Optional[int] = None) -> np.ndarray:
        """Numeric dataset -> float32 array (HDF5 converts the stored type, e.g. uint8 frames).  `start`/`count`
        select rows [start, start+count) along axis 0 (negative start counts from the end, like `ds[-1:]`)."""
        shp = self.shape(name)
        d = _ok(self.lib.H5Dopen2(self.fid, name.strip("/").encode(), H5P_DEFAULT), f"open dataset {name}")
        space = _ok(self.lib.H5Dget_space(d), "dataspace")
        mem = H5S_ALL
        try:
            nd = len(shp)
            if nd and (start != 0 or count is not None):
                if start < 0:
                    start += shp[0]
                count = shp[0] - start if count is None else count
                if start < 0 or count < 0 or start + count > shp[0]:
                    raise IndexError(f"rows [{start}, {start + count}) outside {name} with {shp[0]} rows")
                shp = (count,) + shp[1:]
                st = (hsize_t * nd)(start, *([0] * (nd - 1)))
                ct = (hsize_t * nd)(*shp)
                _ok(self.lib.H5Sselect_hyperslab(space, 0, st, None, ct, None), "hyperslab")      # H5S_SELECT_SET = 0
                mem = _ok(self.lib.H5Screate_simple(nd, ct, None), "memory dataspace")
            out = np.empty(shp, dtype=np.float32)
            if out.size:
                _ok(self.lib.H5Dread(d, self.lib._t_f32, mem, space if mem != H5S_ALL else H5S_ALL, H5P_DEFAULT,
                                     out.ctypes.data_as(C.c_void_p)), f"read {name}")
            return out
        finally:
            if mem != H5S_ALL:
                self.lib.H5Sclose(mem)
            self.lib.H5Sclose(space)
            self.lib.H5Dclose(d)

    def read_attr(self, group: str, name: str, as_int: bool = False):
        g = _ok(self.lib.H5Gopen2(self.fid, group.strip("/").encode(), H5P_DEFAULT), f"open group {group}")
        at = _ok(self.lib.H5Aopen(g, name.encode(), H5P_DEFAULT), f"open attribute {name}")
        try:
            buf = C.c_int64() if as_int else C.c_float()
            _ok(self.lib.H5Aread(at, self.lib._t_i64 if as_int else self.lib._t_f32, C.byref(buf)), f"read attribute {name}")
            return buf.value
        finally:
            self.lib.H5Aclose(at)
            self.lib.H5Gclose(g)


def write_latent_file(path: str, latents: Dict[str, np.ndarray], attrs: Optional[dict] = None) -> None:
    """latents: {"data/<ep>/latent/<key>": array}; attrs land on the `data` group (total / min_z / max_z)."""
    with File(path, "w") as f:
        f.require_group("data")
        for name, arr in latents.items():
            f.write_dataset(name, arr)
        for k, v in (attrs or {}).items():
            f.write_attr("data", k, v)
