"""Known-answer tests pinning the oracle's restatement of the third-party semantics
(SURVEY.md 8c-3, Appendix A).  CPU only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from latent_diffusion_planning_amd import schedule, weights as W
from oracle import np64, torch32


def test_ddpm_tables_closed_form():
    b, a, acp = np64.ddpm_tables(100)
    np.testing.assert_allclose(b[:3], [6.3128e-4, 1.11694e-3, 1.6029e-3], rtol=2e-5)
    np.testing.assert_allclose(b[97:], [0.55537564, 0.74993926, 0.999], rtol=1e-6)
    np.testing.assert_allclose(acp[:2], [0.9993687, 0.9982525], rtol=1e-6)
    np.testing.assert_allclose(acp[97:], [9.7119267e-4, 2.4285716e-4, 2.4285404e-7], rtol=2e-5)
    s = schedule.make_schedule(100)                       # product-side table == oracle table
    assert np.array_equal(s.betas, b) and np.array_equal(s.alphas_cumprod, acp)


def test_step_coefficients_match_oracle_step():
    s = schedule.make_schedule(100)
    rng = np.random.default_rng(0)
    x, eps, z = rng.standard_normal((3, 5)), rng.standard_normal((3, 5)), rng.standard_normal((3, 5))
    co = schedule.step_coefficients(s, 100, schedule.SAMPLER_DDPM).astype(np.float64)
    for i in (0, 1, 57, 98, 99):
        t = 99 - i
        assert co[i, 0] == t
        x0 = np.clip((x - co[i, 2] * eps) * co[i, 1], -1, 1)
        got = co[i, 3] * x0 + co[i, 4] * x + co[i, 5] * eps + co[i, 6] * z
        np.testing.assert_allclose(got, np64.ddpm_step(eps, t, x, z), rtol=0, atol=2e-6 * max(1, co[i, 1] * 1e-3))
    cd = schedule.step_coefficients(s, 50, schedule.SAMPLER_DDIM).astype(np.float64)
    ts = schedule.step_timesteps(100, 50, schedule.SAMPLER_DDIM)
    assert ts[0] == 98 and ts[-1] == 0 and len(ts) == 50
    for i in (0, 10, 49):
        t = int(ts[i])
        x0 = np.clip((x - cd[i, 2] * eps) * cd[i, 1], -1, 1)
        got = cd[i, 3] * x0 + cd[i, 5] * eps
        np.testing.assert_allclose(got, np64.ddim_step(eps, t, t - 2, x), atol=2e-6)
    # DDIM at S=100 visits the DDPM timesteps
    assert np.array_equal(schedule.step_timesteps(100, 100, schedule.SAMPLER_DDIM),
                          schedule.step_timesteps(100, 100, schedule.SAMPLER_DDPM))
    with pytest.raises(ValueError):
        schedule.step_timesteps(100, 50, schedule.SAMPLER_DDPM)


def test_ddpm_last_step_is_deterministic():
    x = np.ones((2, 3)) * 0.3
    eps = np.ones((2, 3)) * 0.1
    a = np64.ddpm_step(eps, 0, x, np.full((2, 3), 1e9))
    b = np64.ddpm_step(eps, 0, x, np.zeros((2, 3)))
    assert np.array_equal(a, b)


def test_add_noise():
    _, _, acp = np64.ddpm_tables()
    x0, n = np.full((2, 4, 3), 2.0), np.full((2, 4, 3), -1.0)
    out = np64.ddpm_add_noise(x0, n, np.array([0, 99]))
    np.testing.assert_allclose(out[0], math.sqrt(acp[0]) * 2 - math.sqrt(1 - acp[0]))
    np.testing.assert_allclose(out[1], math.sqrt(acp[99]) * 2 - math.sqrt(1 - acp[99]))


def test_mish_values():
    x = np.array([-20.0, -1.0, 0.0, 1.0, 20.0])
    ref = np.array([-20 * math.tanh(math.log1p(math.exp(-20))), -math.tanh(math.log1p(math.exp(-1))),
                    0.0, math.tanh(math.log1p(math.e)), 20.0])
    np.testing.assert_allclose(np64.mish(x), ref, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(F.mish(torch.tensor(x)).numpy(), ref, rtol=1e-12, atol=1e-15)


def test_sinusoid_order_and_values():
    e = np64.sinusoidal_pos_emb(np.array([0, 1, 99]), 256)
    assert e.shape == (3, 256)
    np.testing.assert_allclose(e[0, :128], 0.0)          # sin first
    np.testing.assert_allclose(e[0, 128:], 1.0)          # then cos
    np.testing.assert_allclose(e[1, 0], math.sin(1.0), rtol=1e-7)
    np.testing.assert_allclose(e[1, 127], math.sin(1e-4), rtol=2e-6)
    np.testing.assert_allclose(e[2, 128], math.cos(99.0), rtol=1e-7)
    f = np64.fourier_features(np.array([[1.0]]), 256)
    np.testing.assert_allclose(f[0, 0], math.cos(1.0), rtol=1e-7)     # cos first for the IDM
    np.testing.assert_allclose(f[0, 128], math.sin(1.0), rtol=1e-7)
    # product-side tables agree with the oracle's
    np.testing.assert_allclose(schedule.sinusoidal_table(100, 256, False)[[0, 1, 99]], e, atol=1e-7)
    np.testing.assert_allclose(schedule.sinusoidal_table(100, 256, True)[1], f[0], atol=1e-7)


def test_group_norm_constant_and_ramp():
    x = np.full((2, 4, 16), 3.0)
    y = np64.group_norm(x, np.ones(16), np.zeros(16), 8)
    np.testing.assert_allclose(y, 0.0, atol=1e-12)        # var=0 -> (x-mean)*rsqrt(eps) = 0
    x = np.arange(2 * 4 * 16, dtype=np.float64).reshape(2, 4, 16)
    y = np64.group_norm(x, np.ones(16), np.zeros(16), 8)
    g0 = x[0][:, 0:2]
    ref = (g0 - g0.mean()) / math.sqrt(g0.var() + 1e-6)
    np.testing.assert_allclose(y[0][:, 0:2], ref, rtol=1e-9)
    yt = F.group_norm(torch.tensor(x).transpose(1, 2), 8, eps=1e-6).transpose(1, 2).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-9, atol=1e-9)


def test_conv_identity_kernel_and_padding():
    x = np.random.default_rng(1).standard_normal((2, 8, 3))
    k = np.zeros((5, 3, 3))
    k[2] = np.eye(3)
    np.testing.assert_allclose(np64.conv1d(x, k, np.zeros(3), 1, (2, 2)), x)
    k = np.zeros((5, 3, 3))
    k[0] = np.eye(3)                                      # tap 0 reads x[t-2]
    y = np64.conv1d(x, k, np.zeros(3), 1, (2, 2))
    np.testing.assert_allclose(y[:, 2:], x[:, :-2])
    np.testing.assert_allclose(y[:, :2], 0.0)


def test_same_padding_stride2_is_asymmetric():
    assert np64.same_pads(8, 3, 2) == (0, 1)
    assert np64.same_pads(4, 3, 2) == (0, 1)
    assert np64.same_pads(16, 3, 2) == (0, 1)
    assert np64.same_pads(15, 3, 2) == (1, 1)
    # impulse at the last position only reaches the last output through tap 1 (pads (0,1))
    x = np.zeros((1, 8, 1))
    x[0, 7, 0] = 1.0
    k = np.array([10.0, 20.0, 30.0]).reshape(3, 1, 1)
    y = np64.conv1d(x, k, np.zeros(1), 2, np64.same_pads(8, 3, 2))
    np.testing.assert_allclose(y[0, :, 0], [0, 0, 0, 20.0])
    x[:] = 0
    x[0, 0, 0] = 1.0
    y = np64.conv1d(x, k, np.zeros(1), 2, np64.same_pads(8, 3, 2))
    np.testing.assert_allclose(y[0, :, 0], [10.0, 0, 0, 0])


def test_conv_transpose_impulse_and_two_phase_formula():
    k = np.array([1.0, 2.0, 3.0, 4.0]).reshape(4, 1, 1)
    x = np.zeros((1, 4, 1))
    x[0, 1, 0] = 1.0
    y = np64.conv_transpose1d_same_s2(x, k, np.zeros(1))[0, :, 0]
    # out[2q] = x[q-1] K0 + x[q] K2 ; out[2q+1] = x[q] K1 + x[q+1] K3
    np.testing.assert_allclose(y, [0.0, 4.0, 3.0, 2.0, 1.0, 0.0, 0.0, 0.0])
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 4, 3))
    kk = rng.standard_normal((4, 3, 5))
    y = np64.conv_transpose1d_same_s2(x, kk, np.zeros(5))
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
    for q in range(4):
        np.testing.assert_allclose(y[:, 2 * q], xp[:, q] @ kk[0] + xp[:, q + 1] @ kk[2], atol=1e-12)
        np.testing.assert_allclose(y[:, 2 * q + 1], xp[:, q + 1] @ kk[1] + xp[:, q + 2] @ kk[3], atol=1e-12)
    # torch mapping used by oracle/torch32.py
    w = torch.tensor(kk).flip(0).permute(1, 2, 0).contiguous()
    yt = F.conv_transpose1d(torch.tensor(x).transpose(1, 2), w, stride=2, padding=1).transpose(1, 2).numpy()
    np.testing.assert_allclose(y, yt, atol=1e-12)


def test_unet_rejects_t15_accepts_multiples_of_4():
    spec = W.PlannerSpec(5, 5, down_dims=(16, 32, 64))
    p = W.init_planner_params(spec, 0)
    kw = dict(down_dims=spec.down_dims)
    g = np.zeros((1, 5))
    # T=15: 15 -> 8 -> 4 down, 4 -> 8 -> 16 up: the net itself returns T=16 and the
    # reference then fails in scheduler.step on the (15 vs 16) shape mismatch
    e15 = np64.unet_forward(p, np.zeros((1, 15, 5)), 3, g, **kw)
    assert e15.shape == (1, 16, 5)
    with pytest.raises(ValueError):
        np64.ddpm_step(e15, 3, np.zeros((1, 15, 5)), np.zeros((1, 15, 5)))
    for t in (4, 8, 16):
        assert np64.unet_forward(p, np.zeros((1, t, 5)), 3, g, **kw).shape == (1, t, 5)


def test_param_counts_match_survey():
    assert W.count(W.init_planner_params(W.PlannerSpec(25, 25), 0)) == 65_609_497
    assert W.count(W.init_idm_params(W.IDMSpec(25, 7), 0)) == 1_792_007
    enc = W.vae_encoder_shapes(W.VAESpec())
    assert sum(int(np.prod(s)) for s in enc.values()) == 17_297_616


def test_np64_vs_torch_restatements_agree_small():
    """Two independent restatements (explicit loops vs torch functional) must agree."""
    spec = W.PlannerSpec(7, 7, down_dims=(32, 64, 128))
    p = W.init_planner_params(spec, 11)
    rng = np.random.default_rng(5)
    x, g = rng.standard_normal((3, 8, 7)), rng.uniform(-1, 1, (3, 7))
    ks = np.array([0, 50, 99])
    ref = np64.unet_forward(p, x, ks, g, down_dims=spec.down_dims)
    P = torch32.TorchParams(p, dtype=torch.float64)
    got = torch32.unet_forward(P, torch.tensor(x), ks, torch.tensor(g), down_dims=spec.down_dims).numpy()
    np.testing.assert_allclose(got, ref, atol=5e-7)
    P32 = torch32.TorchParams(p)
    got32 = torch32.unet_forward(P32, torch.tensor(x, dtype=torch.float32), ks,
                                 torch.tensor(g, dtype=torch.float32), down_dims=spec.down_dims).numpy()
    np.testing.assert_allclose(got32, ref, atol=2e-5)

    isp = W.IDMSpec(7, 3)
    ip = W.init_idm_params(isp, 12)
    s, a = rng.uniform(-1, 1, (6, 14)), rng.standard_normal((6, 3))
    ref = np64.idm_forward(ip, s, a, 42)
    got = torch32.idm_forward(torch32.TorchParams(ip, dtype=torch.float64), torch.tensor(s),
                              torch.tensor(a), 42).numpy()
    np.testing.assert_allclose(got, ref, atol=5e-7)


def test_vae_restatements_agree():
    vs = W.VAESpec(block_out_channels=(32, 64, 64), norm_num_groups=32)
    shapes = W.vae_shapes(vs)
    p = W.init_from_shapes(shapes, 3)
    rng = np.random.default_rng(6)
    img = rng.uniform(-1, 1, (2, 16, 16, 3))
    ref = np64.vae_encode_mean(p, img, n_blocks=3)
    P = torch32.TorchParams(p, dtype=torch.float64)
    got = torch32.vae_encode_mean(P, torch.tensor(img), n_blocks=3).numpy()
    assert ref.shape == (2, 4, 4, 4)
    np.testing.assert_allclose(got, ref, atol=1e-10)
    z = rng.uniform(-2, 2, (1, 2, 2, 4))
    ref = np64.vae_decode(p, z, n_blocks=3)
    got = torch32.vae_decode(P, torch.tensor(z), n_blocks=3).numpy()
    assert ref.shape == (1, 3, 8, 8)
    np.testing.assert_allclose(got, ref, atol=1e-10)


def test_normalization_roundtrip_and_clip():
    lo, hi = np.array([-0.162, -0.05, 0.728]), np.array([0.068, 0.058, 1.141])
    v = np.array([[0.0, 0.0, 1.0], [1.0, -1.0, 0.0]])
    n = np64.normalize_bounds(v, lo, hi)
    np.testing.assert_allclose(n[0], (v[0] - lo) / (hi - lo) * 2 - 1)
    u = np64.unnormalize_bounds(n, lo, hi)
    np.testing.assert_allclose(u[0], v[0], atol=1e-12)
    np.testing.assert_allclose(u[1], np.clip(v[1], lo, hi))                  # unnormalize clips
    np.testing.assert_allclose(np64.apply_norm(np.array([-3.0, 0.2, 7.0]), dict(clip_min=-1, clip_max=1), False),
                               [-1, 0.2, 1])
    np.testing.assert_allclose(np64.normalize_bounds(np.array([0.0, 255.0]), 0, 255), [-1, 1])
