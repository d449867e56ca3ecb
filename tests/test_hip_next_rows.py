"""GPU tests of the 'next' rows around the hot path: eval harness on the real agent,
bulk latent pre-encoding, pred_horizon 16 (BASELINE config 3), safetensors weight import.  -m gpu."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import torch32
from tests import cfgs
from tests.fake_env import make_env
from tests.util import assert_close, idm_params, planner_params, rng

pytestmark = pytest.mark.gpu


def _agent(data, vae=None, **over):
    from latent_diffusion_planning_amd.agent import LDPAgent
    kw = cfgs.agent_kwargs(data)
    kw.update(over)
    D = 16 + sum(int(np.prod(data["shape_meta"]["all_shapes"][k])) for k in data["lowdim_obs"])
    A = data["shape_meta"]["ac_dim"]
    ag = LDPAgent.create(0, None, data["shape_meta"], vae_params=vae, **kw)
    return ag.replace(planner_state=ag.planner_state.replace(params=planner_params(D=D, G=D * kw["obs_horizon"])),
                      idm_state=ag.idm_state.replace(params=idm_params(D=D, A=A)))


def test_harness_drives_the_real_agent_with_variable_batches():
    from latent_diffusion_planning_amd.harness import run_eval
    data = cfgs.RM_LIFT
    ag = _agent(data)
    dims = {k: int(np.prod(data["shape_meta"]["all_shapes"][k])) for k in data["lowdim_obs"]}
    env = dict(obs_horizon=1, rgb_viz=None,
               env_kwargs=dict(lowdim_obs=data["lowdim_obs"], rgb_obs=data["rgb_obs"], horizon=12, obs_dims=dims))
    logs, _ = run_eval(env, ag, n_rollout=4, n_proc=4, seed=3, eval_rng=7, env_factory=make_env,
                       keep_latent_keys=True, visualize_plan=False)
    assert set(logs) >= {"success", "reward", "horizon", "total_time"} and logs["horizon"] <= 12
    assert logs["policy_calls"] >= 3 and 1.0 <= logs["mean_batch"] <= 4.0
    ag._engine.check_fault()


def test_aloha_harness_drives_the_real_agent_on_raw_frames():
    """utils/aloha_env_utils.py:51-163 with the real LDPAgent (aloha configuration, raw camera frames in): every policy call is
    `sample_viz(dict(obs=obs_dict), rng)` on a (1, 1, 64, 64, 3) frame + (1, 1, 14) qpos -- StableVAE encode, planner, IDM, plan_viz
    decode -- and the reference's `((plan_viz + 1) / 2 * 255).astype(np.uint8).transpose(0, 1, 3, 4, 2)` runs on the returned array."""
    from latent_diffusion_planning_amd.harness import run_aloha_eval
    from tests.fake_env import make_aloha_env
    data = cfgs.ALOHA_CUBE
    ag = _agent(data, vae=W.init_vae_params(seed=2))
    env_params = dict(obs_horizon=1, lowdim_obs=data["lowdim_obs"], rgb_obs=data["rgb_obs"], rgb_viz="top_image",
                      env_kwargs=dict(task_name="sim_transfer_cube", horizon=8))
    # the viz camera is not an agent input: the harness hands the policy only what the env produced for lowdim_obs + rgb_obs
    env_params_policy = dict(env_params, rgb_viz=None)
    logs, videos = run_aloha_eval(env_params_policy, ag, n_rollout=2, seed=1, eval_rng=3, env_factory=make_aloha_env, episode_len=8)
    assert set(logs) >= {"success", "reward", "horizon", "avg_reward", "total_time", "policy_calls"} and logs["horizon"] <= 9
    assert logs["policy_calls"] >= 2 and len(videos) == 2
    ag._engine.check_fault()
    ag._engine.close()


def test_bulk_preencode_matches_direct_encode_and_pads_ragged_tail():
    from latent_diffusion_planning_amd.engine import HipEngine
    from latent_diffusion_planning_amd.preencode import encode_dataset
    vp = W.init_vae_params(seed=2, decoder=False)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(vae=vp)
    g = rng(44)
    eps = {"demo_0": {"agentview_image": g.integers(0, 256, (11, 64, 64, 3)).astype(np.float32)},
           "demo_1": {"agentview_image": g.integers(0, 256, (4, 64, 64, 3)).astype(np.float32)}}
    lat, attrs = encode_dataset(e, eps, shard=4)                 # 11 = 4 + 4 + 3 (padded tail)
    assert lat["data/demo_0/latent/agentview_image"].shape == (11, 2, 2, 4) and attrs["total"] == 2
    assert attrs["min_z"] <= 0.0 <= attrs["max_z"]
    x = torch.tensor(eps["demo_0"]["agentview_image"]) / 255 * 2 - 1
    direct = e.vae_encode(x).cpu().numpy()
    assert_close(lat["data/demo_0/latent/agentview_image"], direct, 2e-6, "sharded vs direct encode")
    P = torch32.TorchParams(vp, dtype=torch.float64)
    ref = torch32.vae_encode_mean(P, x[8:].double()).numpy()
    assert_close(lat["data/demo_0/latent/agentview_image"][8:], ref, 5e-5, "ragged tail vs oracle")
    e.close()


def test_bulk_preencode_file_to_file_writes_the_reference_latent_hdf5(tmp_path):
    """process_sdvae_data.py:55-121 end to end: robomimic image file in (uint8 obs + next_obs), latent.hdf5 out in the
    layout data/robomimic_latent_data.py:94-96 opens; rows = obs frames + the last next_obs frame; values = the
    engine's encode of those frames; attributes total / min_z / max_z."""
    from latent_diffusion_planning_amd import hdf5_io
    from latent_diffusion_planning_amd.engine import HipEngine
    from latent_diffusion_planning_amd.preencode import encode_hdf5
    try:
        hdf5_io.load()
    except hdf5_io.HDF5Unavailable as e:
        pytest.skip(str(e))
    vp = W.init_vae_params(seed=2, decoder=False)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(vae=vp)
    g = rng(45)
    keys = ["agentview_image", "robot0_eye_in_hand_image"]
    frames = {f"demo_{i}": {k: g.integers(0, 256, (n + 1, 64, 64, 3), dtype=np.uint8) for k in keys} for i, n in [(0, 9), (1, 3), (10, 5)]}
    src, dst = str(tmp_path / "image.hdf5"), str(tmp_path / "latent.hdf5")
    with hdf5_io.File(src, "w") as f:
        for ep, obs in frames.items():
            for k, fr in obs.items():
                f.write_dataset(f"data/{ep}/obs/{k}", fr[:-1], np.uint8)
                f.write_dataset(f"data/{ep}/next_obs/{k}", fr[1:], np.uint8)
    attrs = encode_hdf5(e, src, dst, keys, shard=4)
    lo, hi = 0.0, 0.0
    with hdf5_io.File(dst) as f:
        assert f.keys("data") == ["demo_0", "demo_1", "demo_10"] and f.keys("data/demo_1") == ["latent"]
        assert f.read_attr("data", "total", as_int=True) == 3
        for ep, obs in frames.items():
            for k, fr in obs.items():
                z = f.read_dataset(f"data/{ep}/latent/{k}")
                assert z.shape == (fr.shape[0], 2, 2, 4)
                direct = e.vae_encode(torch.tensor(fr.astype(np.float32)) / 255 * 2 - 1).cpu().numpy()
                assert_close(z, direct, 2e-6, f"{ep}/{k} file vs direct encode")
                lo, hi = min(lo, float(z.min())), max(hi, float(z.max()))
        assert f.read_attr("data", "min_z") == np.float32(lo) == np.float32(attrs["min_z"])
        assert f.read_attr("data", "max_z") == np.float32(hi) == np.float32(attrs["max_z"])
    e.close()


@pytest.mark.parametrize("B", [2, 300])
def test_pred_horizon_16(B):
    """BASELINE config 3 read as T=16: levels run at T = 16 / 8 / 4."""
    from latent_diffusion_planning_amd.engine import HipEngine
    pp = planner_params()
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=16, action_horizon=4)
    e.load_params(planner=pp)
    g = rng(160 + B)
    x, cond = g.standard_normal((B, 16, 25)), g.uniform(-1, 1, (B, 25))
    got = e.unet_forward(torch.tensor(x, dtype=torch.float32), 77, torch.tensor(cond, dtype=torch.float32))
    n = min(B, 3)
    P = torch32.TorchParams(pp, dtype=torch.float64)
    ref = torch32.unet_forward(P, torch.tensor(x[:n]), 77, torch.tensor(cond[:n])).numpy()
    assert_close(got[:n].cpu().numpy(), ref, 2e-5, f"unet forward T=16 B={B}")
    plans = e.plan_sample(torch.tensor(cond, dtype=torch.float32), seed=1, sampler="ddim", n_steps=5)
    assert plans.shape == (B, 16, 25) and torch.isfinite(plans).all()
    e.check_fault()
    e.close()


def test_safetensors_checkpoint_roundtrip(tmp_path):
    from latent_diffusion_planning_amd.engine import HipEngine
    pp = planner_params()
    f = str(tmp_path / "ckpt.safetensors")
    W.save_safetensors(f, planner_params=pp)
    back = W.load_safetensors(f)["planner_params"]
    assert list(back) == list(pp) and all(np.array_equal(back[k], pp[k]) for k in pp)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(planner=back)
    g = rng(3)
    x, cond = g.standard_normal((2, 8, 25)), g.uniform(-1, 1, (2, 25))
    got = e.unet_forward(torch.tensor(x, dtype=torch.float32), 5, torch.tensor(cond, dtype=torch.float32))
    P = torch32.TorchParams(pp, dtype=torch.float64)
    ref = torch32.unet_forward(P, torch.tensor(x), 5, torch.tensor(cond)).numpy()
    assert_close(got.cpu().numpy(), ref, 2e-5, "forward from a safetensors checkpoint")
    e.close()
