"""Reference configurations restated as plain dicts (agent/ldp_agent.yaml,
data/cfg/rm_lift/latent_img.yaml:19-61, data/cfg/rm_square/latent_img.yaml:19-61, data/cfg/rm_can/latent_img.yaml:19-64,
data/cfg/aloha_cube/latent_wrist.yaml:19-52).  Constant tables only (names, shapes, normalisation bounds)."""
import numpy as np

AGENT_KW = dict(
    name="ldp_agent",
    planner=dict(diffusion_step_embed_dim=256, down_dims=[256, 512, 1024], kernel_size=5, n_groups=8, downsample=True),
    idm_net=dict(n_blocks=3, dropout_rate=None, use_layer_norm=True, hidden_dim=256),
    preprocess_time=dict(output_size=256, learnable=False),
    cond_encoder=dict(hidden_dims=[256, 256], activations="mish", activate_final=False),
    vae_pretrain_path=None, vae_feature_dim=16, use_planner=True, use_idm=True,
    planner_n_diffusion_steps=100, idm_n_diffusion_steps=100,
    alpha_planner=1, alpha_idm=1, lr=1e-4, end_lr=1e-6, idm_lr=1e-4, idm_end_lr=1e-6,
    warmup_steps=500, decay_steps=100000, update_planner_every=1, update_idm_every=1,
    update_idm_after=-1, update_planner_until=-1, update_planner_after=-1, grad_clip=100,
)

RM_LIFT = dict(
    data_name="rm_lift_latent_img64_data",
    lowdim_obs=["robot0_eef_pos", "robot0_eef_quat", "robot0_gripper_qpos"],
    rgb_obs=["latent_agentview_image"],
    shape_meta=dict(ac_dim=7, all_shapes=dict(robot0_eef_pos=[3], robot0_eef_quat=[4],
                                              robot0_eye_in_hand_image=[64, 64, 3], agentview_image=[64, 64, 3],
                                              robot0_gripper_qpos=[2], optimal=[1]), use_images=True),
    obs_normalization=dict(
        obs=dict(
            robot0_eef_pos=dict(min=[-0.162, -0.05, 0.728], max=[0.068, 0.058, 1.141]),
            robot0_eef_quat=dict(min=[0.847, -0.283, -0.025, -0.065], max=[1.1, 0.364, 0.178, 0.05]),
            robot0_gripper_qpos=dict(min=[0.013, -0.044], max=[0.044, -0.016]),
            agentview_image=dict(min=0, max=255),
            robot0_eye_in_hand_image=dict(min=0, max=255),
            latent_agentview_image=dict(min=-10, max=10),
            latent_robot0_eye_in_hand_image=dict(min=-7, max=7)),
        actions=dict(clip_min=-1, clip_max=1)),
    obs_horizon=1, pred_horizon=8, action_horizon=4,
)

_RM_SHAPES = dict(ac_dim=7, all_shapes=dict(robot0_eef_pos=[3], robot0_eef_quat=[4],
                                            robot0_eye_in_hand_image=[64, 64, 3], agentview_image=[64, 64, 3],
                                            robot0_gripper_qpos=[2], optimal=[1]), use_images=True)

# BASELINE configs[2]: rm_square, read as pred_horizon 16 -- its own normalisation table
RM_SQUARE = dict(
    data_name="rm_square_latent_img64_data",
    lowdim_obs=["robot0_eef_pos", "robot0_eef_quat", "robot0_gripper_qpos"],
    rgb_obs=["latent_agentview_image"],
    shape_meta=_RM_SHAPES,
    obs_normalization=dict(
        obs=dict(
            object=dict(min=[-0.5394, -1.089, 0.0005, -0.798, -0.778, -1.1, -1.1, -1.073, -0.974, -1.26, -1.1, -1.1, -0.94, 0],
                        max=[0.6183, 1.128, 1.265, 1.1, 0.84, 1.1, 1.1, 1.12, 1.25, 0.941, 1.1, 1.1, 0.92, 1.01]),
            robot0_eef_pos=dict(min=[-1.6, -1, 0.62], max=[0.418, 1.01, 1.695]),
            robot0_eef_quat=dict(min=[-0.748, -1.1, -0.7, -0.79], max=[1.1, 1.0814, 0.7665, 0.6346]),
            robot0_gripper_qpos=dict(min=[-0.002, -0.05], max=[0.05, 0.0027]),
            agentview_image=dict(min=0, max=255),
            robot0_eye_in_hand_image=dict(min=0, max=255),
            latent_agentview_image=dict(min=-10, max=10),
            latent_robot0_eye_in_hand_image=dict(min=-10, max=10)),
        actions=dict(clip_min=-1, clip_max=1)),
    obs_horizon=1, pred_horizon=16, action_horizon=4,
)

# BASELINE configs[4]: rm_can (best-of-N candidates, 50-step DDIM)
RM_CAN = dict(
    data_name="rm_can_latent_img64_data",
    lowdim_obs=["robot0_eef_pos", "robot0_eef_quat", "robot0_gripper_qpos"],
    rgb_obs=["latent_agentview_image"],
    shape_meta=_RM_SHAPES,
    obs_normalization=dict(
        obs=dict(
            object=dict(min=[-0.023, -0.461, 0.759, -0.661, -0.614, -0.729, -1.099, -0.115, -0.11, 0.004, -1.1, -1.1, -0.877, 0],
                        max=[0.316, 0.5, 1.293, 0.704, 0.774, 1.1, 1.1, 0.307, 0.378, 0.362, 1.098, 1.1, 0.601, 0.915]),
            robot0_eef_pos=dict(min=[-0.081, -0.465, 0.774], max=[0.326, 0.454, 1.347]),
            robot0_eef_quat=dict(min=[0.532, -0.809, -0.251, -0.377], max=[1.1, 0.52, 0.152, 0.089]),
            robot0_gripper_qpos=dict(min=[0.014, -0.044], max=[0.045, -0.011]),
            agentview_image=dict(min=0, max=255),
            robot0_eye_in_hand_image=dict(min=0, max=255),
            optimal=dict(min=0, max=1),
            latent_agentview_image=dict(min=-10, max=10),
            latent_robot0_eye_in_hand_image=dict(min=-5, max=5)),
        actions=dict(clip_min=-1, clip_max=1)),
    obs_horizon=1, pred_horizon=8, action_horizon=4,
)

ALOHA_CUBE = dict(
    data_name="alohasim_cube_latent_data",
    lowdim_obs=["qpos"], rgb_obs=["latent_wrist64_image"],
    shape_meta=dict(ac_dim=14, all_shapes=dict(qpos=[14], qvel=[14], optimal=[1]), use_images=True),
    obs_normalization=dict(
        obs=dict(
            qpos=dict(min=[-0.01079, -1.7412, 0.65322, -0.01885, -0.90132, -0.00152, 0.08767, -0.49369, -1.74741,
                           -0.0368, -0.72513, -0.33, -1.15083, 0.08986],
                      max=[0.00295, 0.00084, 1.47407, 0.09342, 0.42986, 1.74146, 1.02575, 0.4624, 0.30437, 1.32953,
                           0.75156, 1.17165, 1.09885, 1.07227]),
            wrist64_image=dict(min=0, max=255),
            latent_wrist64_image=dict(min=-5.5, max=5.5),
            optimal=dict(min=0, max=1)),
        actions=dict(min=[-0.01086, -1.74261, 0.65023, -0.01693, -0.91383, -0.00104, 0., -0.49434, -1.74904,
                          -0.04081, -0.72305, -0.33, -1.15162, 0.],
                     max=[0.004, -0.001, 1.47011, 0.09594, 0.42287, 1.74092, 1.1, 0.46257, 0.30254, 1.32924,
                          0.75021, 1.16513, 1.09824, 1.1])),
    obs_horizon=1, pred_horizon=8, action_horizon=4,
)


BY_NAME = {"rm": RM_LIFT, "rm_square": RM_SQUARE, "rm_can": RM_CAN, "aloha": ALOHA_CUBE}


# agent/ldp_hier_agent.yaml: the IDM is a second, two-level ConditionalUnet1D over chunks of idm_horizon actions.  The
# reference's own train_bc.yaml (horizon 16 -> pred_horizon 15, idm_horizon 4) yields 3 planner states, which its
# three-level U-Net cannot process; pred_horizon 32 (8 states) is the smallest well-formed reading with action_horizon 4.
HIER_KW = dict(
    name="ldp_hier_agent",
    planner=dict(diffusion_step_embed_dim=256, down_dims=[256, 512, 1024], kernel_size=5, n_groups=8, downsample=True),
    idm_net=dict(diffusion_step_embed_dim=256, down_dims=[256, 512], kernel_size=5, n_groups=8, downsample=True),
    vae_pretrain_path=None, vae_feature_dim=16, use_planner=True, use_idm=True,
    planner_n_diffusion_steps=100, idm_n_diffusion_steps=100,
    alpha_planner=1, alpha_idm=1, lr=1e-4, end_lr=1e-6, idm_lr=1e-4, idm_end_lr=1e-6,
    warmup_steps=500, decay_steps=100000, idm_horizon=4, update_planner_every=1, update_idm_every=1,
    update_idm_after=-1, update_planner_until=-1, update_planner_after=-1, grad_clip=100,
)


def hier_kwargs(data, pred_horizon=32, action_horizon=4):
    kw = dict(HIER_KW)
    kw.update({k: data[k] for k in ("data_name", "lowdim_obs", "rgb_obs", "obs_normalization", "obs_horizon")})
    kw.update(pred_horizon=pred_horizon, action_horizon=action_horizon)
    return kw


def agent_kwargs(data):
    kw = dict(AGENT_KW)
    kw.update({k: data[k] for k in ("data_name", "lowdim_obs", "rgb_obs", "obs_normalization", "obs_horizon",
                                    "pred_horizon", "action_horizon")})
    return kw


def synth_latent_batch(data, B, H, seed, with_actions=False):
    """Observations as the latent datasets deliver them (pre-encoded latents, raw low-dim)."""
    g = np.random.Generator(np.random.PCG64(seed))
    obs = {}
    for k in data["lowdim_obs"]:
        e = data["obs_normalization"]["obs"][k]
        lo, hi = np.asarray(e["min"], np.float32), np.asarray(e["max"], np.float32)
        obs[k] = g.uniform(lo, hi, size=(B, H, lo.size)).astype(np.float32)
    for k in data["rgb_obs"]:
        e = data["obs_normalization"]["obs"][k]
        obs[k] = g.uniform(0.6 * e["min"], 0.6 * e["max"], size=(B, H, 16)).astype(np.float32)
    batch = {"obs": obs}
    if with_actions:
        batch["actions"] = g.uniform(-1, 1, size=(B, H, data["shape_meta"]["ac_dim"])).astype(np.float32)
    return batch
