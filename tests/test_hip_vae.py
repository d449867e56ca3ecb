"""Parity of the StableVAE kernels (3x3 conv primitive, encode, decode, raw-image agent path)
against the oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import np64, torch32
from tests import cfgs
from tests.util import assert_close, idm_params, planner_params, rng

pytestmark = pytest.mark.gpu
DEFAULT_F16 = 1                            # vae_split_f16: two fp16 planes / three products (0: three bf16 planes / six)


def _f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


@pytest.fixture(scope="module")
def vae_params():
    return W.init_vae_params(seed=2)


@pytest.fixture(scope="module")
def eng(vae_params):
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(vae=vae_params)
    yield e
    e.close()


@pytest.mark.parametrize("S,cin,cout,stride,N", [(64, 128, 128, 1, 1), (32, 128, 256, 1, 2), (16, 256, 256, 1, 3),
                                                 (8, 256, 256, 1, 2), (4, 256, 256, 1, 5), (2, 256, 256, 1, 9),
                                                 (2, 256, 32, 1, 3), (64, 128, 128, 2, 1), (32, 256, 256, 2, 2),
                                                 (16, 256, 256, 2, 2), (8, 256, 256, 2, 3), (4, 256, 256, 2, 7),
                                                 (64, 128, 32, 1, 1), (2, 64, 256, 1, 2)])
def test_conv3x3_primitive(S, cin, cout, stride, N):
    from latent_diffusion_planning_amd.engine import conv2d_3x3
    g = rng(S * 7 + cin + stride)
    x = g.standard_normal((N, S, S, cin))
    k = g.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    b = 0.1 * g.standard_normal(cout)
    pads = ((1, 1), (1, 1)) if stride == 1 else ((0, 1), (0, 1))
    ref = np64.conv2d(x, k, b, stride, pads)
    got = conv2d_3x3(_f32(x).cuda(), k, b, stride).cpu().numpy()
    # (the stride-1 64-column tile is the four-wave one the engine runs: one fmaf chain over K = 9 Cin, no K split over waves -- 1e-5 of max|y| ~ 3)
    scale = max(1.0, float(np.abs(ref).max()))
    assert_close(got / scale, ref / scale, 1e-5, f"conv3x3 S={S} {cin}->{cout} stride {stride}")


@pytest.mark.parametrize("S,cin,cout,N,res,dual", [(64, 128, 128, 1, False, True), (64, 128, 128, 2, True, False),
                                                     (64, 256, 128, 1, True, True), (32, 128, 256, 2, False, True),
                                                     (32, 256, 256, 3, True, True), (16, 256, 256, 3, True, True),
                                                     (16, 16, 128, 8, False, False),
                                                     (64, 128, 128, 2, True, 2), (32, 256, 256, 3, False, 2), (16, 16, 128, 8, True, 2)])
def test_conv3x3_split_operand_primitive(S, cin, cout, N, res, dual):
    """The same convolution on the 16-bit matrix pipe -- three bf16 planes per operand and six plane products (dual 0 / 1: one or two
    accumulators) or two fp16 planes and three products (dual = 2: x = h + l' / 2^11), fp32 accumulate: same tolerance as the exact-fp32
    primitive above; the per-tile column sums it leaves for the next GroupNorm are the sums of what it wrote."""
    from latent_diffusion_planning_amd.engine import conv2d_3x3_split
    g = rng(S * 11 + cin + cout + N)
    x = g.standard_normal((N, S, S, cin))
    x[:, :, :, ::7] *= 30.0                                  # mixed magnitudes inside a contraction
    k = g.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    b = 0.1 * g.standard_normal(cout)
    r = g.standard_normal((N, S, S, cout)) if res else None
    ref = np64.conv2d(np.asarray(_f32(x).numpy(), np.float64), np.asarray(_f32(k).numpy(), np.float64), b, 1,
                      ((1, 1), (1, 1)))
    if res:
        ref = ref + np.asarray(_f32(r).numpy(), np.float64)
    got, st = conv2d_3x3_split(_f32(x).cuda(), k, b, _f32(r).cuda() if res else None, dual=dual, with_stats=True)
    got = got.cpu().numpy()
    scale = float(np.abs(ref).max())
    assert_close(got / scale, ref / scale, 3e-6, f"split conv3x3 S={S} {cin}->{cout} (relative to max|y| = {scale:.1f})")
    tiles = got.reshape(N * S * S // 256, 256, cout).astype(np.float64)
    st = st.cpu().numpy()
    assert_close(st[:, :, 0] / 256, tiles.sum(1) / 256, 1e-5 * scale, "column sums")
    assert_close(st[:, :, 1] / 256, (tiles ** 2).sum(1) / 256, 1e-5 * scale * scale, "column sums of squares")


def test_vae_split_operand_convs_agree_with_the_exact_fp32_ones(eng):
    """Option vae_split = 0 puts every 3x3 conv back on v_mfma_f32_16x16x4_f32; the two encoders agree to fp32 round-off."""
    img = _f32(rng(77).uniform(-1, 1, (3, 64, 64, 3)))
    a = eng.vae_encode(img).cpu().numpy()
    eng.set_option("vae_split", 0)
    try:
        b = eng.vae_encode(img).cpu().numpy()
    finally:
        eng.set_option("vae_split", 1)
    assert not np.array_equal(a, b), "the option did not switch the conv path"
    assert_close(a, b, 2e-5, "split-operand vs exact-fp32 encoder")


def test_the_product_library_refuses_timing_ablations(eng):
    """The `dbg` / `repeat` switches (results wrong by construction) are compiled into libldp_hip_abl.so only."""
    from latent_diffusion_planning_amd._lib import LDPHipError
    with pytest.raises(LDPHipError, match="libldp_hip_abl"):
        eng.set_option("dbg", 8)
    with pytest.raises(LDPHipError, match="libldp_hip_abl"):
        eng.set_option("repeat", 2)
    eng.set_option("dbg", 0)
    eng.set_option("repeat", 1)
    assert eng.active_debug_options() == ""


def test_vae_split_operand_margins(eng, vae_params):
    """Error of the encoder against the float64 oracle with the 3x3 convs on exact-fp32 MFMA, on split operands with
    two accumulators and with one: the split forms must not use more of the 5e-5 budget than twice the fp32 one
    (tools/split_bf16_probe.hip measures them at or below the fp32 chain's error).  The numbers go to
    gpurun_out/r4/vae_margins.json when that directory exists (-> profiles/r04_split_vae_margins.json)."""
    import json, os
    img = rng(4242).uniform(-1, 1, (4, 64, 64, 3))
    P = torch32.TorchParams(vae_params, dtype=torch.float64)
    ref = torch32.vae_encode_mean(P, torch.tensor(img)).numpy()
    z = rng(4243).uniform(-3, 3, (2, 2, 2, 4))
    refd = torch32.vae_decode(P, torch.tensor(z)).numpy()
    out = {}
    try:
        for tag, opts in (("fp32_mfma", dict(vae_split=0, vae_split_f16=0)),
                          ("split6_single", dict(vae_split=1, vae_split_f16=0)),
                          ("f16x3", dict(vae_split=1, vae_split_f16=1))):
            for k, v in opts.items():
                eng.set_option(k, v)
            e = float(np.abs(eng.vae_encode(_f32(img)).cpu().numpy() - ref).max())
            d = float(np.abs(eng.vae_decode(_f32(z)).cpu().numpy() - refd).max())
            out[tag] = dict(encode_max_abs_err=e, decode_max_abs_err=d)
    finally:
        eng.set_option("vae_split", 1)
        eng.set_option("vae_split_f16", DEFAULT_F16)
    print(json.dumps(out))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r4")
    if os.path.isdir(d):
        json.dump(out, open(os.path.join(d, "vae_margins.json"), "w"), indent=1)
    for tag in ("split6_single", "f16x3"):
        assert out[tag]["encode_max_abs_err"] <= max(2 * out["fp32_mfma"]["encode_max_abs_err"], 5e-6), out
        assert out[tag]["decode_max_abs_err"] <= max(2 * out["fp32_mfma"]["decode_max_abs_err"], 1e-5), out
        assert out[tag]["encode_max_abs_err"] < 5e-5 and out[tag]["decode_max_abs_err"] < 1e-4, out


@pytest.mark.parametrize("N", [1, 3])
def test_vae_encode_matches_oracle(eng, vae_params, N):
    g = rng(900 + N)
    img = g.uniform(-1, 1, (N, 64, 64, 3))
    P = torch32.TorchParams(vae_params, dtype=torch.float64)
    ref = torch32.vae_encode_mean(P, torch.tensor(img)).numpy()
    got = eng.vae_encode(_f32(img)).cpu().numpy()
    assert got.shape == (N, 2, 2, 4)
    assert_close(got, ref, 5e-5, f"vae encode N={N}")


def test_vae_encode_matches_np64_definition(eng, vae_params):
    img = rng(5).uniform(-1, 1, (1, 64, 64, 3))
    ref = np64.vae_encode_mean(vae_params, img)
    assert_close(eng.vae_encode(_f32(img)).cpu().numpy(), ref, 5e-5, "vae encode vs np64")


def test_vae_decode_matches_oracle(eng, vae_params):
    z = rng(6).uniform(-3, 3, (2, 2, 2, 4))
    P = torch32.TorchParams(vae_params, dtype=torch.float64)
    ref = torch32.vae_decode(P, torch.tensor(z)).numpy()
    got = eng.vae_decode(_f32(z)).cpu().numpy()
    assert got.shape == (2, 3, 64, 64)
    assert_close(got, ref, 1e-4, "vae decode")


def test_agent_raw_image_path(vae_params):
    """sample_viz on raw [0,255] images: normalise -> VAE encode -> latent normalise -> planner ->
    IDM, and plan_viz through the decoder; a short DDIM schedule keeps the oracle cheap."""
    from latent_diffusion_planning_amd.agent import LDPAgent
    data = cfgs.RM_LIFT
    pp, ip = planner_params(), idm_params()
    ag = LDPAgent.create(0, None, data["shape_meta"], vae_params=vae_params, **cfgs.agent_kwargs(data))
    ag = ag.replace(planner_state=ag.planner_state.replace(params=pp), idm_state=ag.idm_state.replace(params=ip))
    B, D, A, S = 2, 25, 7, 10
    g = rng(321)
    low = cfgs.synth_latent_batch(data, B, 1, 9)["obs"]
    obs = {k: v for k, v in low.items() if not k.startswith("latent_")}
    obs["agentview_image"] = g.integers(0, 256, (B, 1, 64, 64, 3)).astype(np.float32)
    batch = {"obs": obs}
    noise = dict(x_init=g.standard_normal((B, 8, D)), a_init=g.standard_normal((B * 4, A)))
    act, met = ag.sample_viz(batch, 0, noise={k: _f32(v) for k, v in noise.items()}, decode=True,
                             sampler="ddim", n_steps=S)

    # oracle with the float64 torch restatement plugged in for the loops
    def pfn(params, obs_cond, x_init, step_noise, n_train, n_steps, sampler):
        P = torch32.TorchParams(params, dtype=torch.float64)
        return torch32.planner_sample(P, torch.tensor(obs_cond), torch.tensor(x_init), None, n_train=n_train,
                                      n_steps=n_steps, sampler=sampler).numpy()

    def ifn(params, trans, a_init, step_noise, n_train, n_steps, sampler):
        P = torch32.TorchParams(params, dtype=torch.float64)
        return torch32.idm_sample(P, torch.tensor(trans), torch.tensor(a_init), None, n_train=n_train,
                                  n_steps=n_steps, sampler=sampler).numpy()

    orc = np64.AgentOracle(dict(ag.config), pp, ip, np64.to64(vae_params), data["obs_normalization"], pfn, ifn)
    ref_a, ref_m = orc.sample_viz(batch, noise["x_init"], None, noise["a_init"], None, decode=True,
                                  sampler="ddim", n_steps=S)
    assert met["plan_viz"].shape == (B, 5, 3, 64, 64)
    assert_close(np.array(met["plan"]), ref_m["plan"], 1e-4, "plan (raw images)")
    assert_close(np.array(act), ref_a, 1e-4, "action (raw images; rm actions are clipped, not scaled)")
    # the harness idiom, verbatim (utils/rm_env_utils.py:185-186)
    plan_viz = met["plan_viz"]
    pv8 = (np.clip((np.array(plan_viz) + 1)/2, 0, 1) * 255).astype(np.uint8)
    assert pv8.shape == (B, 5, 3, 64, 64)
    assert_close(np.array(plan_viz), ref_m["plan_viz"], 5e-4, "plan_viz")
    # encode alone, through the agent method (normalised latent, (h, w, c) flattening)
    enc = ag.vae_encode(ag._postprocess(batch)["obs"])
    ref_enc = orc.vae_encode(orc.postprocess(batch)["obs"])
    assert_close(enc["latent_agentview_image"].cpu().numpy(), ref_enc["latent_agentview_image"], 2e-5, "vae_encode")


@pytest.mark.parametrize("S,LC", [(128, 4), (64, 8), (96, 4)])
def test_other_latent_shapes_match_oracle(S, LC):
    """agent/ldp_agent.py:69-80 lists vae_feature_dim 32 (2x2x8 latent), 36 (3x3x4 latent of 96x96 frames: levels of 96 / 48 / 24 / 12 / 6 / 3
    pixels, the last on the 3-pixel conv tile of round 5) and 64 (4x4x4 latent of 128x128 frames) besides the shipped 16: the same
    kernels, another image side / latent width."""
    from latent_diffusion_planning_amd.engine import HipEngine
    vp = W.init_vae_params(W.VAESpec(latent_channels=LC), seed=2)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4, image_size=S,
                  vae_latent_channels=LC)
    e.load_params(vae=vp)
    g = rng(S + LC)
    img = g.uniform(-1, 1, (2, S, S, 3))
    P = torch32.TorchParams(vp, dtype=torch.float64)
    ref = torch32.vae_encode_mean(P, torch.tensor(img), latent_channels=LC).numpy()
    assert_close(e.vae_encode(_f32(img)).cpu().numpy(), ref, 5e-5, f"encode {S}x{S}, {LC} latent channels")
    z = g.uniform(-3, 3, (2, S // 32, S // 32, LC))
    assert_close(e.vae_decode(_f32(z)).cpu().numpy(), torch32.vae_decode(P, torch.tensor(z)).numpy(), 1e-4,
                 f"decode to {S}x{S}")
    e.close()


@pytest.mark.parametrize("fd,side,LC", [(32, 2, 8), (64, 4, 4), (36, 3, 4)])
def test_agent_with_other_latent_shapes(fd, side, LC):
    """vae_feature_dim 32 (2x2x8 latent, obs_dim 41) and 64 (4x4x4 latent of 128x128 frames, obs_dim 73): raw
    frames in, plan_viz out; a short DDIM schedule keeps the oracle cheap."""
    from latent_diffusion_planning_amd.agent import LDPAgent
    data = cfgs.RM_LIFT
    S_img = 32 * side
    D, A, B, S = 9 + fd, 7, 2, 5
    meta = dict(data["shape_meta"], all_shapes=dict(data["shape_meta"]["all_shapes"], agentview_image=[S_img, S_img, 3]))
    kw = dict(cfgs.agent_kwargs(data), vae_feature_dim=fd)
    vp = W.init_vae_params(W.VAESpec(latent_channels=LC), seed=2)
    pp, ip = planner_params(D=D), idm_params(D=D, A=A)
    ag = LDPAgent.create(0, None, meta, vae_params=vp, **kw)
    ag = ag.replace(planner_state=ag.planner_state.replace(params=pp), idm_state=ag.idm_state.replace(params=ip))
    assert ag.config["obs_dim"] == D
    g = rng(3200 + fd)
    low = cfgs.synth_latent_batch(data, B, 1, 9)["obs"]
    obs = {k: v for k, v in low.items() if not k.startswith("latent_")}
    obs["agentview_image"] = g.integers(0, 256, (B, 1, S_img, S_img, 3)).astype(np.float32)
    batch = {"obs": obs}
    noise = dict(x_init=g.standard_normal((B, 8, D)), a_init=g.standard_normal((B * 4, A)))
    act, met = ag.sample_viz(batch, 0, noise={k: _f32(v) for k, v in noise.items()}, sampler="ddim", n_steps=S)

    def pfn(params, obs_cond, x_init, step_noise, n_train, n_steps, sampler):
        Pp = torch32.TorchParams(params, dtype=torch.float64)
        return torch32.planner_sample(Pp, torch.tensor(obs_cond), torch.tensor(x_init), None, n_train=n_train,
                                      n_steps=n_steps, sampler=sampler).numpy()

    def ifn(params, trans, a_init, step_noise, n_train, n_steps, sampler):
        Pi = torch32.TorchParams(params, dtype=torch.float64)
        return torch32.idm_sample(Pi, torch.tensor(trans), torch.tensor(a_init), None, n_train=n_train,
                                  n_steps=n_steps, sampler=sampler).numpy()

    Pv = torch32.TorchParams(vp, dtype=torch.float64)

    class Orc(np64.AgentOracle):                              # the float64 torch VAE in place of the NumPy loops
        def vae_encode(self, o):
            new = {}
            for key, v in o.items():
                if f"latent_{key}" not in self.cfg["rgb_obs"]:
                    new[key] = np.asarray(v, np.float64)
                    continue
                z = torch32.vae_encode_mean(Pv, torch.tensor(np.asarray(v, np.float64).reshape((-1,) + v.shape[-3:])),
                                            latent_channels=LC).numpy()
                new[f"latent_{key}"] = np64.apply_norm(z.reshape(v.shape[0], v.shape[1], -1),
                                                       self.norm["obs"][f"latent_{key}"], True)
            return new

        def vae_decode(self, feats):
            b, hh = feats.shape[:2]
            z = np.asarray(feats, np.float64)[:, :, :fd].reshape(b * hh, side, side, LC)      # agent/ldp_agent.py:69-80
            z = np64.apply_norm(z, self.norm["obs"][self.cfg["rgb_obs"][0]], False)
            img = torch32.vae_decode(Pv, torch.tensor(z)).numpy()
            return img.reshape((b, hh) + img.shape[1:])
    orc = Orc(dict(ag.config), pp, ip, None, data["obs_normalization"], pfn, ifn)
    ref_a, ref_m = orc.sample_viz(batch, noise["x_init"], None, noise["a_init"], None, decode=True, sampler="ddim",
                                  n_steps=S)
    assert met["plan"].shape == (B, 5, D) and met["plan_viz"].shape == (B, 5, 3, S_img, S_img)
    assert_close(np.array(met["plan"]), ref_m["plan"], 1e-4, f"plan (vae_feature_dim {fd})")
    assert_close(np.array(act), ref_a, 1e-4, f"action (vae_feature_dim {fd})")
    assert_close(np.array(met["plan_viz"]), ref_m["plan_viz"], 5e-4, f"plan_viz (vae_feature_dim {fd})")
    ag._engine.close()


@pytest.mark.parametrize("N", [512, 2048])
def test_vae_encode_at_shard_size(eng, vae_params, N):
    """BASELINE configs[3] (aloha: StableVAE 64x64 encode, batch 2048 over 4 GPUs) at the per-GPU shard size and at
    the whole batch on one GPU: the 64x64 128-channel convs then run 65 536-131 072 row tiles (`blockIdx.z` folding
    in launch_one, 32-bit element offsets in the epilogue: 2048 x 64 x 64 x 128 = 2^30 elements per tensor).
    Size-independent properties: finite, no fault, and the latents of rows {0, 1, N-2, N-1} are those of the same
    frames encoded as a 16-frame batch -- BITWISE: an image never depends on its neighbours, whatever the grid.
    Two of the rows are also checked against the float64 definition (reference: agent/ldp_agent.py:46-64,
    process_sdvae_data.py:95-110)."""
    g = rng(4000 + N)
    base = g.uniform(-1, 1, (16, 64, 64, 3)).astype(np.float32)
    img = torch.tensor(base, device="cuda").repeat(N // 16, 1, 1, 1)
    # every image distinct: add a small per-image offset pattern (keeps [-1, 1])
    scale = torch.linspace(0.5, 1.0, N, device="cuda").reshape(N, 1, 1, 1)
    img = (img * scale).contiguous()
    out = eng.vae_encode(img)
    eng.check_fault()
    assert out.shape == (N, 2, 2, 4) and torch.isfinite(out).all()
    rows = [0, 1, N - 2, N - 1]
    pick = torch.cat([img[rows], img[2:14]]).contiguous()                 # the four rows inside a 16-frame batch
    small = eng.vae_encode(pick)
    assert torch.equal(out[rows], small[:4]), "rows of the large batch differ from the same frames in a 16-frame batch"
    ref = np64.vae_encode_mean(vae_params, img[[0, N - 1]].double().cpu().numpy())
    assert_close(out[[0, N - 1]].cpu().numpy(), ref, 5e-5, f"vae encode rows 0 and {N - 1} of {N}")


def test_aloha_agent_on_raw_frames_at_shard_size(vae_params):
    """configs[3] end to end at the per-GPU shard: LDPAgent.sample on 512 raw wrist64_image frames (normalise ->
    StableVAE encode -> latent normalise -> DDPM-100 planner + IDM, one joint graph).  Finite, inside the action
    bounds, no fault, the last 480 rows bit-identical to the same rows sampled as their own batch (same launch
    regime -- 353..512 plans since round 4 --, same Philox rows), and the encoder's part bit-identical to a 16-frame batch."""
    ag, data = None, cfgs.ALOHA_CUBE
    from tests.util import make_agent
    ag, data = make_agent("aloha", planner_params(D=30), idm_params(D=30, A=14), vae=vae_params)
    B = 512
    g = rng(5120)
    low = cfgs.synth_latent_batch(data, B, 1, 3)["obs"]
    obs = {k: v for k, v in low.items() if not k.startswith("latent_")}
    obs["wrist64_image"] = g.integers(0, 256, (B, 1, 64, 64, 3)).astype(np.float32)
    batch = {"obs": obs}
    act, met = ag.sample(batch, 11)
    a, p = np.array(act), np.array(met["plan"])
    ag._engine.check_fault()
    assert a.shape == (B, 4, 14) and p.shape == (B, 5, 30) and np.isfinite(a).all() and np.isfinite(p).all()
    lo, hi = (np.asarray(data["obs_normalization"]["actions"][k], np.float32) for k in ("min", "max"))
    assert (a >= lo - 1e-5).all() and (a <= hi + 1e-5).all()
    tail = {"obs": {k: v[32:] for k, v in obs.items()}}
    act2, met2 = ag.sample(tail, 11, row_offset=32)
    assert np.array_equal(np.array(met2["plan"]), p[32:]), "plans depend on the batch they were sampled in"
    assert_close(np.array(act2), a[32:], 1e-5, "actions of the tail as its own batch")     # IDM split may differ by row count
    enc = ag.vae_encode(ag._postprocess(batch)["obs"])["latent_wrist64_image"]
    enc16 = ag.vae_encode(ag._postprocess({"obs": {k: v[:16] for k, v in obs.items()}})["obs"])["latent_wrist64_image"]
    assert np.array_equal(np.array(enc)[:16], np.array(enc16))
    ag._engine.close()
