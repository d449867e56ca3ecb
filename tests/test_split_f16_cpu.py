"""CPU checks of the two-plane fp16 split arithmetic the default split tiles compute with (tconv SPLIT = 3 / 4, sconv3 NPL = 2):
x ~ h + l' / 2^11 with h = fp16(x), l' = fp16((x - h) * 2^11); products h h + 2^-11 (h l' + l' h), every fp16 x fp16 product exact in fp32.
The emulation (tests/split_emulate.py) is test infrastructure like the oracle; nothing here touches the product library."""
import numpy as np
import torch

from tests.split_emulate import H16_SCALE, SplitF16, planes_f16


def test_two_fp16_planes_carry_22_bits_whatever_the_magnitude():
    g = np.random.Generator(np.random.PCG64(7))
    # magnitudes from fp16's subnormal range to the activations' largest values; the scaled low plane keeps its 11 bits everywhere
    x = torch.tensor(g.standard_normal(200000) * np.exp(g.uniform(np.log(1e-6), np.log(3e3), 200000)), dtype=torch.float32)
    h, l = planes_f16(x)
    assert torch.isfinite(h).all() and torch.isfinite(l).all()
    rec = h.double() + l.double() / H16_SCALE
    rel = ((rec - x.double()).abs() / x.double().abs()).numpy()
    big = x.abs().numpy() >= 2.0 ** -14                       # h is a normal fp16 there
    assert rel[big].max() <= 2.0 ** -22, rel[big].max()       # 11 + 11 bits, both planes rounded to nearest: 2^-23 typical, 2^-22 worst
    # below fp16's normal range h loses bits, the residue still lands in l': the absolute error stays below 2^-25 * 2^-11 * (1 + ...) ~ 3e-11
    assert (rec - x.double()).abs().numpy()[~big].max() <= 4e-11
    # what the kernels rely on: both planes are exactly representable in fp16 (the conversions in split4h / pack_* are exact round trips)
    assert torch.equal(h, h.to(torch.float16).to(torch.float32)) and torch.equal(l, l.to(torch.float16).to(torch.float32))


def test_three_product_contraction_is_at_the_fp32_round_off_level():
    """K = 5120 (the planner's longest contraction: 5 taps x 1024 channels), N(0,1) operands with mixed magnitudes: the three-product form with
    float32 accumulation against float64, next to the plain float32 contraction of the same data."""
    g = np.random.Generator(np.random.PCG64(8))
    a = torch.tensor(g.standard_normal((64, 5120)) * np.exp(g.uniform(-3, 3, (1, 5120))), dtype=torch.float32)
    w = torch.tensor(g.standard_normal((128, 5120)) / np.sqrt(5120), dtype=torch.float32)
    ref = a.double() @ w.double().T
    scale = (a.double().abs() @ w.double().abs().T)             # sum |a b|: the yardstick of profiles/r04_split_probe.txt
    ulp = 2.0 ** -24
    e32 = (((a @ w.T).double() - ref).abs() / scale).max().item() / ulp
    f = SplitF16("acc32")
    e16 = ((f.linear(a, w).double() - ref).abs() / scale).max().item() / ulp
    f64 = SplitF16("acc64")
    e16w = ((f64.linear(a, w).double() - ref).abs() / scale).max().item() / ulp
    assert e16w <= 0.6, e16w                                     # what the arithmetic drops (representation + l' l'): well under one fp32 ulp of sum |a b|
    assert e16 <= max(2.0 * e32, 4.0), (e16, e32)                # with fp32 accumulation: the fp32 contraction's own level
