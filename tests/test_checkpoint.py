"""Reference checkpoints (orbax PyTreeCheckpointer aggregate files, flax msgpack wire format) -> parameter trees.
CPU only.  The wire format is pinned by a byte string assembled by hand from flax/serialization.py's definition
(no orbax here to write one); everything else round-trips through it."""
import os
import struct

import numpy as np
import pytest

msgpack = pytest.importorskip("msgpack")

from latent_diffusion_planning_amd import checkpoint as ck
from latent_diffusion_planning_amd import weights as W


def _hand_made():
    """{'idm_params': {'Dense_0': {'kernel': f32[[1,2],[3,4]], 'bias': f32[5,6]}}, 'cfg': {'seed': 7}, 'data': {'step': i32 scalar 3}}
    byte for byte: fixmap / fixstr / ext8 headers written out, payload = fixarray(shape fixarray, dtype fixstr, bin8)."""
    def fixstr(s):
        b = s.encode()
        assert len(b) < 32
        return bytes([0xa0 | len(b)]) + b

    def ext(code, shape, dtype, raw):
        payload = bytes([0x93]) + bytes([0x90 | len(shape)]) + bytes(shape) + fixstr(dtype) + bytes([0xc4, len(raw)]) + raw
        return bytes([0xc7, len(payload), code]) + payload

    kernel = ext(1, (2, 2), "float32", struct.pack("<4f", 1, 2, 3, 4))
    bias = ext(1, (2,), "float32", struct.pack("<2f", 5, 6))
    step = ext(3, (), "int32", struct.pack("<i", 3))
    dense = bytes([0x82]) + fixstr("kernel") + kernel + fixstr("bias") + bias
    idm = bytes([0x81]) + fixstr("Dense_0") + dense
    cfg = bytes([0x81]) + fixstr("seed") + bytes([0x07])
    data = bytes([0x81]) + fixstr("step") + step
    return bytes([0x83]) + fixstr("idm_params") + idm + fixstr("cfg") + cfg + fixstr("data") + data


def test_reads_the_flax_msgpack_wire_format(tmp_path):
    d = tmp_path / "100.ckpt"
    d.mkdir()
    (d / "checkpoint").write_bytes(_hand_made())
    raw = ck.restore(str(d))
    assert raw["cfg"] == {"seed": 7} and raw["data"]["step"] == 3 and raw["data"]["step"].dtype == np.int32
    k = raw["idm_params"]["Dense_0"]["kernel"]
    assert k.dtype == np.float32 and k.tolist() == [[1, 2], [3, 4]] and k.flags.writeable
    trees = ck.param_trees(raw)
    assert list(trees) == ["idm_params"] and list(trees["idm_params"]) == ["Dense_0/kernel", "Dense_0/bias"]
    assert ck.restore(str(d / "checkpoint"))["cfg"] == {"seed": 7}              # the file itself works too
    # and the writer emits exactly these bytes for this tree (same key order, same headers)
    out = ck.save(str(tmp_path / "w.ckpt"), {"idm_params": {"Dense_0": {"kernel": k, "bias": raw["idm_params"]["Dense_0"]["bias"]}},
                                              "cfg": {"seed": 7}, "data": {"step": np.int32(3)}})
    assert open(out, "rb").read() == _hand_made()


def test_full_parameter_trees_round_trip_and_select(tmp_path):
    pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
    ip = W.init_idm_params(W.IDMSpec(25, 7), 1)
    vp = W.init_vae_params(seed=2, decoder=False)
    path = str(tmp_path / "5000.ckpt")
    ck.save(path, {"planner_params": pp, "idm_params": ip, "vae_params": vp, "planner_ema_params": pp,
                   "cfg": {"agent": {"name": "ldp"}}, "data": {"actions": np.zeros((2, 7), np.float32)}})
    raw = ck.restore(path)
    assert set(raw) == {"planner_params", "idm_params", "vae_params", "planner_ema_params", "cfg", "data"}
    assert all(isinstance(v, dict) for v in raw["planner_params"].values())      # nested on disk, like agent.get_params()
    trees = ck.param_trees(raw)
    assert set(trees) == {"planner_params", "idm_params", "vae_params"}           # ema skipped (train_bc.py:230-232)
    for name, ref in (("planner_params", pp), ("idm_params", ip), ("vae_params", vp)):
        assert list(trees[name]) == list(ref)
        for k in ref:
            np.testing.assert_array_equal(trees[name][k], ref[k])
    assert set(ck.param_trees(raw, ["idm_params"])) == {"idm_params"}              # cfg.restore_keys
    W.check_params(trees["planner_params"], W.planner_shapes(W.PlannerSpec(25, 25)))
    W.check_params(trees["idm_params"], W.idm_shapes(W.IDMSpec(25, 7)))


def test_chunked_arrays_bfloat16_and_placeholders(tmp_path):
    a = np.arange(24, dtype=np.float32).reshape(4, 6)
    ext = lambda x: msgpack.ExtType(1, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes()), use_bin_type=True))
    bf = (np.array([1.0, -2.5, 0.15625], np.float32).view(np.uint32) >> 16).astype(np.uint16)
    tree = {"planner_params": {"big": {"__msgpack_chunked_array__": True, "shape": [4, 6],
                                       "chunks": {"0": ext(a.reshape(-1)[:10]), "1": ext(a.reshape(-1)[10:])}},
                               "half": msgpack.ExtType(1, msgpack.packb(([3], "bfloat16", bf.tobytes()), use_bin_type=True))},
            "idm_params": {"Dense_0": {"kernel": "PLACEHOLDER://idm_params.Dense_0.kernel"}}}
    d = tmp_path / "7.ckpt"
    d.mkdir()
    (d / "checkpoint").write_bytes(msgpack.packb(tree, use_bin_type=True))
    raw = ck.restore(str(d))
    np.testing.assert_array_equal(raw["planner_params"]["big"], a)
    np.testing.assert_array_equal(raw["planner_params"]["half"], np.array([1.0, -2.5, 0.15625], np.float32))
    assert set(ck.param_trees(raw, ["planner_params"])) == {"planner_params"}
    with pytest.raises(ck.CheckpointError, match="outside the aggregate file"):
        ck.param_trees(raw)


def test_errors_name_the_problem(tmp_path):
    with pytest.raises(ck.CheckpointError, match="no such checkpoint"):
        ck.restore(str(tmp_path / "nope.ckpt"))
    d = tmp_path / "ts.ckpt"
    (d / "planner_params.kernel").mkdir(parents=True)
    with pytest.raises(ck.CheckpointError, match="no aggregate file"):
        ck.restore(str(d))
    m = tmp_path / "new_orbax.ckpt"
    m.mkdir()
    (m / "_METADATA").write_text("{}")
    with pytest.raises(ck.CheckpointError, match="_METADATA / OCDBT layout.*orbax-checkpoint 0.5.14"):
        ck.restore(str(m))
    (tmp_path / "junk").write_bytes(b"\xc1not msgpack")
    with pytest.raises(ck.CheckpointError, match="not a flax/orbax msgpack"):
        ck.restore(str(tmp_path / "junk"))


class _State:
    def __init__(self, params):
        self.params, self.ema_params = params, params

    def replace(self, **kw):
        s = _State(self.params)
        s.__dict__.update(kw)
        return s


class _Agent:
    """The three things load_snapshot touches: `<prefix>_state`, `vae_params`, `.replace` (LDPAgent has them; needs no GPU here)."""
    def __init__(self):
        self.planner_state, self.idm_state, self.vae_params = _State({"w": np.zeros(1)}), _State({"w": np.zeros(1)}), None

    def replace(self, **kw):
        a = _Agent()
        a.__dict__.update(self.__dict__)
        a.__dict__.update(kw)
        return a


def test_load_snapshot_follows_train_bc(tmp_path):
    ip = W.init_idm_params(W.IDMSpec(25, 7), 1)
    vp = W.init_vae_params(seed=2, decoder=False)
    path = str(tmp_path / "9.ckpt")
    ck.save(path, {"idm_params": ip, "idm_ema_params": ip, "vae_params": vp, "encoder_params": {"shared_params": {"w": np.ones(2, np.float32)}}})
    old = _Agent()
    new = ck.load_snapshot(old, path)
    assert new is not old and old.idm_state.params == {"w": old.idm_state.params["w"]}
    assert list(new.idm_state.params) == list(ip) and new.idm_state.ema_params is new.idm_state.params
    assert list(new.vae_params) == list(vp) and list(new.planner_state.params) == ["w"]      # not in the file: untouched
    only = ck.load_snapshot(old, path, restore_keys=["vae_params"])
    assert only.idm_state is old.idm_state and list(only.vae_params) == list(vp)
    with pytest.raises(ck.CheckpointError, match="no \\*_params tree selected"):
        ck.load_snapshot(old, path, restore_keys=["cfg"])


def test_vae_pretrain_path_containers_are_told_apart_by_suffix_first(tmp_path):
    """'runs/5000.ckpt/vae.npz' and 'ckpts/vae.safetensors' contain 'ckpt' but are not orbax files."""
    from latent_diffusion_planning_amd import checkpoint, weights as W
    from latent_diffusion_planning_amd.agent import load_pretrained_vae
    tree = {"encoder/conv_in/bias": np.arange(4, dtype=np.float32), "quant_conv/bias": np.ones(2, np.float32)}
    d = tmp_path / "5000.ckpt"
    d.mkdir()
    W.save_npz(str(d / "vae.npz"), vae_params=tree)
    got = load_pretrained_vae(str(d / "vae.npz"))
    assert set(got) == set(tree) and np.array_equal(got["encoder/conv_in/bias"], tree["encoder/conv_in/bias"])
    c = tmp_path / "ckpts"
    c.mkdir()
    W.save_safetensors(str(c / "vae.safetensors"), vae_params=tree)
    got = load_pretrained_vae(str(c / "vae.safetensors"))
    assert np.array_equal(got["quant_conv/bias"], tree["quant_conv/bias"])
    # the reference's own rule still holds for a real checkpoint path
    checkpoint.save(str(tmp_path / "100.ckpt"), {"vae_params": W.unflatten(tree)})
    got = load_pretrained_vae(str(tmp_path / "100.ckpt"))
    assert np.array_equal(np.asarray(got["encoder"]["conv_in"]["bias"] if "encoder" in got else got["encoder/conv_in/bias"]), tree["encoder/conv_in/bias"])
