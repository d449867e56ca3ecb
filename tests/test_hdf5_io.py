"""The latent.hdf5 container (hdf5_io.py: the HDF5 C library through ctypes) -- CPU only.  The independent reader is
h5py where the image has one (the conda python3.9 beside the main interpreter): what the reference's data layer does
with the file (data/robomimic_latent_data.py:94-96) is done with it, in a subprocess."""
import os
import subprocess

import numpy as np
import pytest

from latent_diffusion_planning_amd import hdf5_io, preencode

try:
    hdf5_io.load()
except hdf5_io.HDF5Unavailable as _e:            # pragma: no cover
    pytest.skip(str(_e), allow_module_level=True)


def _latents():
    g = np.random.default_rng(5)
    lat = {f"data/demo_{i}/latent/{k}": g.standard_normal((n, 12, 12, 4)).astype(np.float32)
           for i, n in [(0, 6), (1, 1), (12, 3)] for k in ("agentview_image", "robot0_eye_in_hand_image")}
    return lat, {"total": 3, "min_z": float(min(v.min() for v in lat.values())), "max_z": float(max(v.max() for v in lat.values()))}


def test_latent_file_round_trip(tmp_path):
    lat, attrs = _latents()
    p = str(tmp_path / "latent.hdf5")
    preencode.save_latents(p, lat, attrs)                                       # *.hdf5 -> the HDF5 container
    with open(p, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    with hdf5_io.File(p) as f:
        assert f.keys("data") == ["demo_0", "demo_1", "demo_12"]
        assert f.keys("data/demo_12/latent") == ["agentview_image", "robot0_eye_in_hand_image"]
        for k, v in lat.items():
            assert f.exists(k) and f.shape(k) == v.shape
            np.testing.assert_array_equal(f.read_dataset(k), v)
        assert not f.exists("data/demo_3/latent/agentview_image")
        assert f.read_attr("data", "total", as_int=True) == 3
        assert f.read_attr("data", "min_z") == np.float32(attrs["min_z"]) and f.read_attr("data", "max_z") == np.float32(attrs["max_z"])
    assert preencode.demo_order(["demo_10", "demo_2", "demo_0"]) == ["demo_0", "demo_2", "demo_10"]


def test_image_file_rows_and_types(tmp_path):
    """The input side of the bulk pre-encode: uint8 frames come back as float32 values, `next_obs[-1]` is one row."""
    g = np.random.default_rng(6)
    fr = g.integers(0, 256, (7, 8, 8, 3), dtype=np.uint8)
    p = str(tmp_path / "image.hdf5")
    with hdf5_io.File(p, "w") as f:
        f.write_dataset("data/demo_0/obs/cam", fr[:-1], np.uint8)
        f.write_dataset("data/demo_0/next_obs/cam", fr[1:], np.uint8)
        f.write_dataset("data/demo_0/empty", np.zeros((0, 4), np.float32))
        with pytest.raises(ValueError):
            f.write_dataset("data/demo_0/actions", np.zeros(3), np.float64)
        with pytest.raises(hdf5_io.HDF5Error):
            f.write_dataset("data/demo_0/obs/cam", fr, np.uint8)                 # exists already
    with hdf5_io.File(p) as f:
        np.testing.assert_array_equal(f.read_dataset("data/demo_0/obs/cam"), fr[:-1].astype(np.float32))
        np.testing.assert_array_equal(f.read_dataset("data/demo_0/next_obs/cam", -1), fr[-1:].astype(np.float32))
        np.testing.assert_array_equal(f.read_dataset("data/demo_0/obs/cam", 2, 3), fr[2:5].astype(np.float32))
        assert f.read_dataset("data/demo_0/empty").shape == (0, 4)
        with pytest.raises(IndexError):
            f.read_dataset("data/demo_0/obs/cam", 5, 3)
        with pytest.raises(hdf5_io.HDF5Error):
            f.read_dataset("data/demo_0/obs/none")
    with pytest.raises(hdf5_io.HDF5Error):
        hdf5_io.File(str(tmp_path / "missing.hdf5"))


H5PY_PYTHON = "/opt/conda/bin/python3.9"
_READER = r"""
import sys, json, numpy as np, h5py
f = h5py.File(sys.argv[1], 'r')
demos = list(f['data'].keys())
demos = [demos[i] for i in np.argsort([int(e[5:]) for e in demos])]
out = {'demos': demos, 'attrs': {k: [str(np.asarray(v).dtype), float(v)] for k, v in f['data'].attrs.items()}, 'sets': {}}
for d in demos:
    for k in f['data'][d]['latent']:
        a = f['data'][d]['latent'][k][:]
        out['sets']['data/%s/latent/%s' % (d, k)] = [str(a.dtype), list(a.shape), a.astype(np.float64).sum(), float(a.flat[-1])]
print(json.dumps(out))
"""


def _has_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")
def test_h5py_reads_the_file_the_way_the_data_layer_does(tmp_path):
    import json
    lat, attrs = _latents()
    p = str(tmp_path / "latent.hdf5")
    preencode.save_latents(p, lat, attrs, fmt="hdf5")
    r = subprocess.run([H5PY_PYTHON, "-c", _READER, p], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    assert got["demos"] == ["demo_0", "demo_1", "demo_12"]
    assert got["attrs"]["total"] == ["int64", 3.0]
    assert got["attrs"]["min_z"] == ["float32", float(np.float32(attrs["min_z"]))]
    assert got["attrs"]["max_z"] == ["float32", float(np.float32(attrs["max_z"]))]
    assert set(got["sets"]) == set(lat)
    for k, v in lat.items():
        dt, shp, tot, last = got["sets"][k]
        assert dt == "float32" and tuple(shp) == v.shape
        assert tot == float(v.astype(np.float64).sum()) and last == float(v.flat[-1])


def test_npz_container_still_there(tmp_path):
    lat, attrs = _latents()
    p = str(tmp_path / "latent.npz")
    preencode.save_latents(p, lat, attrs)
    z = np.load(p)
    assert int(z["data.attrs/total"]) == 3 and set(lat) <= set(z.files)
    with pytest.raises(ValueError):
        preencode.save_latents(p, lat, attrs, fmt="zarr")
