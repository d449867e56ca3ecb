"""Algorithmic work of the hot path: 2*MACs of *non-padding* taps only, so a
kernel that multiplies zero padding is not credited.  Used by bench.py for `roofline.achieved`.
Reference shapes: networks/diffusion_nets_v2.py:113-169, networks/mlp_diffusion_nets.py:32-68,
model/stable_vae_model.yaml:4-16."""
from __future__ import annotations

from .weights import IDMSpec, PlannerSpec, VAESpec

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X, v_mfma_f32_16x16x4_f32 / 32x32x2 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def _valid_taps_same(t: int, k: int) -> int:
    """sum over output positions of the taps of a k-wide, stride-1, pad k//2 conv that hit data."""
    p = k // 2
    return sum(1 for to in range(t) for j in range(k) if 0 <= to + j - p < t)


def _valid_taps_down(t_in: int) -> int:
    """k=3 stride-2 conv with XLA SAME pads (0,1) for even t_in."""
    return sum(1 for q in range(t_in // 2) for j in range(3) if 2 * q + j < t_in)


def _valid_taps_up(t_in: int) -> int:
    """transposed k=4 stride-2: out[2q]=x[q-1]K0+x[q]K2, out[2q+1]=x[q]K1+x[q+1]K3."""
    n = 0
    for q in range(t_in):
        n += (q - 1 >= 0) + 1 + 1 + (q + 1 < t_in)
    return n


def planner_forward_flops(spec: PlannerSpec, T: int, hoisted: bool = False) -> float:
    """FLOPs of one ConditionalUnet1D evaluation for one sample.  hoisted=True leaves out the
    k-only / plan-only work this build moves out of the loop (time MLP + FiLM Dense)."""
    k = spec.kernel_size
    macs = 0.0
    film = 0.0
    t = T
    blocks = spec.blocks()
    L = len(spec.down_dims)
    bi = 0

    def block(cin, cout, proj, t):
        m = _valid_taps_same(t, k) * (cin * cout + cout * cout)
        if proj:
            m += t * cin * cout
        return m

    for lvl in range(L):
        for _ in range(2):
            cin, cout, proj = blocks[bi]
            macs += block(cin, cout, proj, t)
            film += spec.cond_dim * 2 * cout
            bi += 1
        if lvl < L - 1:
            c = spec.down_dims[lvl]
            macs += _valid_taps_down(t) * c * c
            t //= 2
    for _ in range(2):
        cin, cout, proj = blocks[bi]
        macs += block(cin, cout, proj, t)
        film += spec.cond_dim * 2 * cout
        bi += 1
    for lvl in range(L - 1):
        for _ in range(2):
            cin, cout, proj = blocks[bi]
            macs += block(cin, cout, proj, t)
            film += spec.cond_dim * 2 * cout
            bi += 1
        c = list(reversed(spec.down_dims[:-1]))[lvl]
        macs += _valid_taps_up(t) * c * c
        t *= 2
    c0 = spec.down_dims[0]
    macs += _valid_taps_same(t, k) * c0 * c0 + t * c0 * spec.input_dim
    e = spec.diffusion_step_embed_dim
    time_mlp = 2 * e * 4 * e
    total = macs if hoisted else macs + film + time_mlp
    return 2.0 * total


def idm_forward_flops(spec: IDMSpec) -> float:
    """FLOPs of one MLPDiffusion evaluation for one row."""
    h = spec.hidden_dim
    macs = spec.time_dim * spec.cond_hidden[0] + spec.cond_hidden[0] * spec.cond_hidden[1]
    macs += spec.in_dim * h + spec.n_blocks * (h * 4 * h * 2) + h * spec.action_dim
    return 2.0 * macs


def plan_flops(pspec: PlannerSpec, T: int, n_steps: int, ispec: IDMSpec = None, idm_rows: int = 0,
               idm_steps: int = 0) -> float:
    f = n_steps * planner_forward_flops(pspec, T)
    if ispec is not None:
        f += idm_steps * idm_rows * idm_forward_flops(ispec)
    return f
