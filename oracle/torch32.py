"""ORACLE (test infrastructure, NOT a product path) -- independent float32 torch-CPU
restatement of the same hot path as oracle/np64.py, written against torch.nn.functional
primitives with the Flax->torch layout / padding / flip mappings derived in SURVEY.md
Appendix A.  It exists (i) to catch transcription errors in np64.py (the two must agree to
float32 round-off) and (ii) as the timed "CPU port" baseline in bench.py (`cpu_baseline`,
kind "port": proxy for the unavailable JAX-CPU reference path).

PARITY UNPINNED (see oracle/np64.py header): no executable reference, no golden vectors.

Mappings used (each has a KAT in tests/test_oracle_kats.py):
  * flax Conv kernel (k, Cin, Cout)           -> F.conv1d weight (Cout, Cin, k) = permute(2,1,0)
  * flax Conv stride-2 'SAME' (k=3, even T)   -> F.pad(x, (0, 1)) + conv1d(stride=2)
  * flax ConvTranspose(k=4, s=2, SAME, transpose_kernel=False)
        -> F.conv_transpose1d(x, weight[ci, co, j] = kernel[3-j, ci, co], stride=2, padding=1)
  * flax GroupNorm / LayerNorm eps = 1e-6
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


class TorchParams:
    """Flat flax-path dict -> torch tensors, with the layout conversions cached."""

    def __init__(self, params, dtype=torch.float32, device="cpu"):
        self.raw = params
        self.dtype, self.device = dtype, device
        self._cache = {}

    def t(self, key):
        v = self._cache.get(key)
        if v is None:
            v = torch.as_tensor(np.asarray(self.raw[key]), dtype=self.dtype, device=self.device)
            self._cache[key] = v
        return v

    def conv1d_w(self, key):            # (k, Cin, Cout) -> (Cout, Cin, k)
        ck = ("c1", key)
        if ck not in self._cache:
            self._cache[ck] = self.t(key).permute(2, 1, 0).contiguous()
        return self._cache[ck]

    def convT_w(self, key):             # (k, Cin, Cout) -> (Cin, Cout, k) flipped along k
        ck = ("ct", key)
        if ck not in self._cache:
            self._cache[ck] = self.t(key).flip(0).permute(1, 2, 0).contiguous()
        return self._cache[ck]

    def conv2d_w(self, key):            # (kh, kw, Cin, Cout) -> (Cout, Cin, kh, kw)
        ck = ("c2", key)
        if ck not in self._cache:
            self._cache[ck] = self.t(key).permute(3, 2, 0, 1).contiguous()
        return self._cache[ck]

    def has(self, key):
        return key in self.raw


def _freqs(dim, device):
    half = dim // 2
    step = np.float32(np.log(np.float32(10000.0))) / np.float32(half - 1)
    f = np.exp(np.arange(half, dtype=np.float32) * -step).astype(np.float32)
    return torch.as_tensor(f, device=device)


def _ksteps(k, n, device):
    if torch.is_tensor(k):
        return k.to(device=device, dtype=torch.float32).reshape(-1).expand(n) if k.numel() == 1 \
            else k.to(device=device, dtype=torch.float32).reshape(n)
    k = np.asarray(k)
    return torch.as_tensor(np.broadcast_to(k.reshape(-1) if k.ndim else k, (n,)).astype(np.float32),
                           device=device)


# ------------------------------------------------------------------------------ planner
def _conv_block(P: TorchParams, x, prefix, groups, k):
    y = F.conv1d(x, P.conv1d_w(f"{prefix}/Conv_0/kernel"), P.t(f"{prefix}/Conv_0/bias"),
                 padding=k // 2)
    y = F.group_norm(y, groups, P.t(f"{prefix}/GroupNorm_0/scale"),
                     P.t(f"{prefix}/GroupNorm_0/bias"), eps=1e-6)
    return F.mish(y)


def _res_block(P, x, cond_mish, prefix, groups, k, proj):
    out = _conv_block(P, x, f"{prefix}/Conv1dBlock_0", groups, k)
    emb = F.linear(cond_mish, P.t(f"{prefix}/Dense_0/kernel").t(), P.t(f"{prefix}/Dense_0/bias"))
    c = out.shape[1]
    out = emb[:, :c, None] * out + emb[:, c:, None]
    out = _conv_block(P, out, f"{prefix}/Conv1dBlock_1", groups, k)
    res = x
    if proj:
        res = F.conv1d(x, P.conv1d_w(f"{prefix}/Conv_0/kernel"), P.t(f"{prefix}/Conv_0/bias"))
    return out + res


def unet_forward(P: TorchParams, x_btc, k, global_cond, down_dims=(256, 512, 1024),
                 kernel_size=5, n_groups=8, embed_dim=256, downsample=True):
    """Channels-first inside; takes / returns (B, T, D) like the reference."""
    dev = x_btc.device
    b = x_btc.shape[0]
    kk = _ksteps(k, b, dev)
    arg = kk[:, None] * _freqs(embed_dim, dev)[None, :]
    e = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1).to(P.dtype)
    e = F.mish(F.linear(e, P.t("Dense_0/kernel").t(), P.t("Dense_0/bias")))
    e = F.linear(e, P.t("Dense_1/kernel").t(), P.t("Dense_1/bias"))
    g = torch.cat([e, global_cond.to(P.dtype)], dim=-1) if global_cond is not None else e
    gm = F.mish(g)
    x = x_btc.to(P.dtype).transpose(1, 2)
    h = []
    idx = 0
    nl = len(down_dims)
    for lvl in range(nl):
        for proj in (True, False):
            x = _res_block(P, x, gm, f"ConditionalResidualBlock1D_{idx}", n_groups, kernel_size, proj)
            idx += 1
        h.append(x)
        if downsample and lvl < nl - 1:
            t = x.shape[-1]
            if t % 2 == 0:
                xp = F.pad(x, (0, 1))
            else:
                xp = F.pad(x, (1, 1))
            x = F.conv1d(xp, P.conv1d_w(f"Downsample1d_{lvl}/Conv_0/kernel"),
                         P.t(f"Downsample1d_{lvl}/Conv_0/bias"), stride=2)
    for _ in range(2):
        x = _res_block(P, x, gm, f"ConditionalResidualBlock1D_{idx}", n_groups, kernel_size, False)
        idx += 1
    for lvl in range(nl - 1):
        x = torch.cat([x, h.pop()], dim=1)
        for proj in (True, False):
            x = _res_block(P, x, gm, f"ConditionalResidualBlock1D_{idx}", n_groups, kernel_size, proj)
            idx += 1
        if downsample:
            x = F.conv_transpose1d(x, P.convT_w(f"Upsample1d_{lvl}/ConvTranspose_0/kernel"),
                                   P.t(f"Upsample1d_{lvl}/ConvTranspose_0/bias"),
                                   stride=2, padding=1)
    x = _conv_block(P, x, "Conv1dBlock_0", 8, kernel_size)
    x = F.conv1d(x, P.conv1d_w("Conv_0/kernel"), P.t("Conv_0/bias"))
    return x.transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------ IDM
def idm_forward(P: TorchParams, s, a, k, time_dim=256, n_blocks=3):
    dev = s.device
    r = s.shape[0]
    kk = _ksteps(k, r, dev)
    arg = kk[:, None] * _freqs(time_dim, dev)[None, :]
    tff = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1).to(P.dtype)
    c = F.mish(F.linear(tff, P.t("MLP_0/Dense_0/kernel").t(), P.t("MLP_0/Dense_0/bias")))
    c = F.linear(c, P.t("MLP_0/Dense_1/kernel").t(), P.t("MLP_0/Dense_1/bias"))
    h = F.linear(torch.cat([a.to(P.dtype), s.to(P.dtype), c], dim=-1),
                 P.t("MLPResNet_0/Dense_0/kernel").t(), P.t("MLPResNet_0/Dense_0/bias"))
    for i in range(n_blocks):
        p = f"MLPResNet_0/MLPResNetBlock_{i}"
        y = F.layer_norm(h, (h.shape[-1],), P.t(f"{p}/LayerNorm_0/scale"),
                         P.t(f"{p}/LayerNorm_0/bias"), eps=1e-6)
        y = F.relu(F.linear(y, P.t(f"{p}/Dense_0/kernel").t(), P.t(f"{p}/Dense_0/bias")))
        h = h + F.linear(y, P.t(f"{p}/Dense_1/kernel").t(), P.t(f"{p}/Dense_1/bias"))
    return F.linear(F.relu(h), P.t("MLPResNet_0/Dense_1/kernel").t(), P.t("MLPResNet_0/Dense_1/bias"))


# ------------------------------------------------------------------------------ schedulers
def _tables(n):
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = torch.tensor([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)],
                         dtype=torch.float32)
    alphas = 1.0 - betas
    return betas, alphas, torch.cumprod(alphas, 0)


def ddpm_step(eps, t, x, noise, tables):
    """float32 arithmetic throughout, like the reference's traced step."""
    betas, alphas, acp = tables
    a_t = acp[t]
    a_prev = acp[t - 1] if t > 0 else torch.tensor(1.0)
    x0 = ((x - (1 - a_t).sqrt() * eps) / a_t.sqrt()).clamp(-1.0, 1.0)
    c0 = a_prev.sqrt() * betas[t] / (1 - a_t)
    cx = alphas[t].sqrt() * (1 - a_prev) / (1 - a_t)
    out = c0 * x0 + cx * x
    if t > 0:
        var = ((1 - a_prev) / (1 - a_t) * betas[t]).clamp(min=1e-20)
        out = out + var.sqrt() * noise
    return out


def ddim_step(eps, t, t_prev, x, tables):
    _, _, acp = tables
    a_t = acp[t]
    a_prev = acp[t_prev] if t_prev >= 0 else torch.tensor(1.0)
    x0 = ((x - (1 - a_t).sqrt() * eps) / a_t.sqrt()).clamp(-1.0, 1.0)
    return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps


@torch.no_grad()
def planner_sample(P, obs_cond, x_init, step_noise=None, n_train=100, n_steps=100,
                   sampler="ddpm", stop_after=None, **kw):
    """stop_after: run only the first `stop_after` of the n_steps steps (bench.py's bounded CPU
    timing; every step costs the same).  None = the whole loop."""
    tables = _tables(n_train)
    x = x_init.to(P.dtype)
    stride = n_train // n_steps
    for i in range(n_steps if stop_after is None else min(stop_after, n_steps)):
        k = (n_steps - 1 - i) * stride
        eps = unet_forward(P, x, k, obs_cond, **kw)
        if sampler == "ddpm":
            x = ddpm_step(eps, k, x, step_noise[i] if k > 0 else None, tables)
        else:
            x = ddim_step(eps, k, k - stride, x, tables)
    return x


@torch.no_grad()
def idm_sample(P, transition, a_init, step_noise=None, n_train=100, n_steps=100, sampler="ddpm",
               stop_after=None):
    tables = _tables(n_train)
    a = a_init.to(P.dtype)
    stride = n_train // n_steps
    for i in range(n_steps if stop_after is None else min(stop_after, n_steps)):
        k = (n_steps - 1 - i) * stride
        eps = idm_forward(P, transition, a, k)
        if sampler == "ddpm":
            a = ddpm_step(eps, k, a, step_noise[i] if k > 0 else None, tables)
        else:
            a = ddim_step(eps, k, k - stride, a, tables)
    return a


# ------------------------------------------------------------------------------ StableVAE
def _gn2(P, x, prefix, groups=32):
    return F.group_norm(x, groups, P.t(f"{prefix}/scale"), P.t(f"{prefix}/bias"), eps=1e-6)


def _resnet2d(P, x, prefix):
    h = F.conv2d(F.silu(_gn2(P, x, f"{prefix}/norm1")), P.conv2d_w(f"{prefix}/conv1/kernel"),
                 P.t(f"{prefix}/conv1/bias"), padding=1)
    h = F.conv2d(F.silu(_gn2(P, h, f"{prefix}/norm2")), P.conv2d_w(f"{prefix}/conv2/kernel"),
                 P.t(f"{prefix}/conv2/bias"), padding=1)
    if P.has(f"{prefix}/conv_shortcut/kernel"):
        x = F.conv2d(x, P.conv2d_w(f"{prefix}/conv_shortcut/kernel"),
                     P.t(f"{prefix}/conv_shortcut/bias"))
    return h + x


def _attn(P, x, prefix):
    n, c, hh, ww = x.shape
    r = _gn2(P, x, f"{prefix}/group_norm").reshape(n, c, hh * ww).transpose(1, 2)
    lin = lambda t, nm: F.linear(t, P.t(f"{prefix}/{nm}/kernel").t(), P.t(f"{prefix}/{nm}/bias"))
    q, k, v = lin(r, "query"), lin(r, "key"), lin(r, "value")
    sc = c ** -0.25
    w = torch.softmax((q * sc) @ (k * sc).transpose(1, 2), dim=-1)
    o = lin(w @ v, "proj_attn")
    return o.transpose(1, 2).reshape(n, c, hh, ww) + x


def _mid(P, x, prefix):
    x = _resnet2d(P, x, f"{prefix}/resnets_0")
    x = _attn(P, x, f"{prefix}/attentions_0")
    return _resnet2d(P, x, f"{prefix}/resnets_1")


@torch.no_grad()
def vae_encode_mean(P, img_nhwc, n_blocks=6, layers=2, latent_channels=4):
    x = img_nhwc.to(P.dtype).permute(0, 3, 1, 2)
    x = F.conv2d(x, P.conv2d_w("encoder/conv_in/kernel"), P.t("encoder/conv_in/bias"), padding=1)
    for i in range(n_blocks):
        for j in range(layers):
            x = _resnet2d(P, x, f"encoder/down_blocks_{i}/resnets_{j}")
        if i != n_blocks - 1:
            p = f"encoder/down_blocks_{i}/downsamplers_0/conv"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), P.conv2d_w(f"{p}/kernel"), P.t(f"{p}/bias"), stride=2)
    x = _mid(P, x, "encoder/mid_block")
    x = F.silu(_gn2(P, x, "encoder/conv_norm_out"))
    x = F.conv2d(x, P.conv2d_w("encoder/conv_out/kernel"), P.t("encoder/conv_out/bias"), padding=1)
    x = F.conv2d(x, P.conv2d_w("quant_conv/kernel"), P.t("quant_conv/bias"))
    return x[:, :latent_channels].permute(0, 2, 3, 1).contiguous()


@torch.no_grad()
def vae_decode(P, z_nhwc, n_blocks=6, layers=2):
    x = z_nhwc.to(P.dtype).permute(0, 3, 1, 2)
    x = F.conv2d(x, P.conv2d_w("post_quant_conv/kernel"), P.t("post_quant_conv/bias"))
    x = F.conv2d(x, P.conv2d_w("decoder/conv_in/kernel"), P.t("decoder/conv_in/bias"), padding=1)
    x = _mid(P, x, "decoder/mid_block")
    for i in range(n_blocks):
        for j in range(layers + 1):
            x = _resnet2d(P, x, f"decoder/up_blocks_{i}/resnets_{j}")
        if i != n_blocks - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            p = f"decoder/up_blocks_{i}/upsamplers_0/conv"
            x = F.conv2d(x, P.conv2d_w(f"{p}/kernel"), P.t(f"{p}/bias"), padding=1)
    x = F.silu(_gn2(P, x, "decoder/conv_norm_out"))
    return F.conv2d(x, P.conv2d_w("decoder/conv_out/kernel"), P.t("decoder/conv_out/bias"), padding=1)
