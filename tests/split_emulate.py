#!/usr/bin/env python3
"""Stage 0(a) of the round-4 split-operand question, on the CPU (no GPU needed): how far do the 100-step
recurrences drift when every contraction of the planner / IDM / StableVAE (conv1d, transposed conv, conv2d,
Dense) is computed from three bf16 planes per operand (x = h + m + l, exact) with N of the 9 plane products
kept?  Every bf16 x bf16 product is exact in fp32, so the model of the matrix pipe is "exact products,
accumulate":

  acc64   the kept products are summed in float64 and rounded to float32 once per output: isolates what the
          DROPPED products cost (N = 9 is then the exact fp32 contraction, correctly rounded);
  acc32   every kept plane product is its own float32 contraction (oneDNN's fp32 accumulation), the N partial
          results added small to large in float32: N fp32 accumulations instead of one -- a pessimistic stand-in
          for an fp32 accumulator that sees N times as many additions.

Everything else (GroupNorm, Mish, FiLM, scheduler step) is float32 as in oracle/torch32.py.  Errors are the max
|difference| to the committed float64 goldens, next to the plain float32 run of the same code ("fp32") -- the
noise floor the exact-fp32 HIP path lives at.  Kill criterion: > 5e-5 on any golden.

  python tests/split_emulate.py [--quick] [--json out.json]
Reads tests/golden and the oracle: test infrastructure (it lives under tests/ for that reason), never the product library.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import numpy as np
import torch
import torch.nn.functional as TF

from oracle import torch32
from tests.cases import load_case, vae_params
from tests.util import idm_params, planner_params

PAIRS9 = [(2, 2), (1, 2), (2, 1), (1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)]   # small to large
PAIRS = {9: PAIRS9, 6: PAIRS9[3:], 3: PAIRS9[6:], 1: PAIRS9[8:]}


def planes(x):
    """fp32 tensor -> (h, m, l) fp32 tensors holding bf16 values, h + m + l == x exactly."""
    h = x.to(torch.bfloat16).to(torch.float32)
    r = x - h
    m = r.to(torch.bfloat16).to(torch.float32)
    r2 = r - m
    l = r2.to(torch.bfloat16).to(torch.float32)
    assert torch.equal(h + m + l, x) or not torch.isfinite(x).all() or (x.abs().min() < 1e-30)
    return h, m, l


H16_SCALE = 2048.0      # the low fp16 plane is stored as (x - h) * 2^11: same magnitude as h, so it keeps its 11 bits whatever |x|


def planes_f16(x):
    """fp32 tensor -> (h, l') fp32 tensors holding fp16 values: h = fp16(x), l' = fp16((x - h) * 2^11); x ~ h + l' / 2^11 to 2^-23 |x|."""
    h = x.to(torch.float16).to(torch.float32)
    l = ((x - h) * H16_SCALE).to(torch.float16).to(torch.float32)
    return h, l


class SplitF16:
    """Two fp16 planes per operand, three products: hh into one accumulator, h l' + l' h into a second one that is added with weight 2^-11
    at the end (every fp16 x fp16 product is exact in fp32; the l' l' term, 2^-22 |ab|, is dropped).  acc as in SplitF."""

    def __init__(self, acc):
        self.acc = acc

    def __getattr__(self, name):
        return getattr(TF, name)

    def _bilinear(self, op, x, w, bias, channel_dim):
        if x.dtype != torch.float32:
            return op(x, w, bias)
        (xh, xl), (wh, wl) = planes_f16(x), planes_f16(w)
        if self.acc == "acc64":
            tot = op(xh.double(), wh.double(), None) + (op(xh.double(), wl.double(), None) + op(xl.double(), wh.double(), None)) / H16_SCALE
        else:
            tot = op(xh, wh, None) + (op(xh, wl, None) + op(xl, wh, None)) * (1.0 / H16_SCALE)
        if bias is not None:
            shape = [1] * tot.dim()
            shape[channel_dim] = -1
            tot = tot + bias.to(tot.dtype).reshape(shape)
        return tot.float()

    linear = lambda self, x, w, bias=None: self._bilinear(lambda a, b, c: TF.linear(a, b, c), x, w, bias, -1)
    conv1d = lambda self, x, w, bias=None, stride=1, padding=0: self._bilinear(lambda a, b, c: TF.conv1d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)
    conv_transpose1d = lambda self, x, w, bias=None, stride=1, padding=0: self._bilinear(lambda a, b, c: TF.conv_transpose1d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)
    conv2d = lambda self, x, w, bias=None, stride=1, padding=0: self._bilinear(lambda a, b, c: TF.conv2d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)


class SplitF:
    """Stands in for torch.nn.functional inside oracle.torch32: the four bilinear ops go through the plane
    products, everything else is torch's."""

    def __init__(self, nprod, acc):
        self.pairs, self.acc = PAIRS[nprod], acc

    def __getattr__(self, name):
        return getattr(TF, name)

    def _bilinear(self, op, x, w, bias, channel_dim):
        if x.dtype != torch.float32:
            return op(x, w, bias)
        xp, wp = planes(x), planes(w)
        wide = self.acc == "acc64"
        tot = None
        for i, j in self.pairs:
            t = op(xp[i].double(), wp[j].double(), None) if wide else op(xp[i], wp[j], None)
            tot = t if tot is None else tot + t
        if bias is not None:
            shape = [1] * tot.dim()
            shape[channel_dim] = -1
            tot = tot + bias.to(tot.dtype).reshape(shape)
        return tot.float()

    def linear(self, x, w, bias=None):
        return self._bilinear(lambda a, b, c: TF.linear(a, b, c), x, w, bias, -1)

    def conv1d(self, x, w, bias=None, stride=1, padding=0):
        return self._bilinear(lambda a, b, c: TF.conv1d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)

    def conv_transpose1d(self, x, w, bias=None, stride=1, padding=0):
        return self._bilinear(lambda a, b, c: TF.conv_transpose1d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)

    def conv2d(self, x, w, bias=None, stride=1, padding=0):
        return self._bilinear(lambda a, b, c: TF.conv2d(a, b, c, stride=stride, padding=padding), x, w, bias, 1)


def f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def run_cases(quick):
    out = {}
    P = torch32.TorchParams(planner_params(), dtype=torch.float32)
    names = [("planner_loop_ddpm100", "ddpm", 100), ("planner_loop_ddim100", "ddim", 100), ("planner_loop_ddim50", "ddim", 50),
             ("planner_loop_t16_ddpm100", "ddpm", 100)]
    if quick:
        names = names[2:3]
    for name, smp, n in names:
        inp, exp = load_case(name)
        out[name] = lambda inp=inp, exp=exp, smp=smp, n=n: err(
            torch32.planner_sample(P, f32(inp["cond"]), f32(inp["x0"]), f32(inp["nz"]) if smp == "ddpm" else None,
                                   n_steps=n, sampler=smp).numpy(), exp["plan"])
    PI = torch32.TorchParams(idm_params(), dtype=torch.float32)
    for smp, n in (("ddpm", 100), ("ddim", 50)):
        inp, exp = load_case(f"idm_loop_rm_{smp}{n}")
        out[f"idm_loop_rm_{smp}{n}"] = lambda inp=inp, exp=exp, smp=smp, n=n: err(
            torch32.idm_sample(PI, f32(inp["tr"]), f32(inp["a0"]), f32(inp["nz"]) if smp == "ddpm" else None,
                               n_steps=n, sampler=smp).numpy(), exp["act"])
    # StableVAE encode: the latent of the aloha raw-image golden (frames normalised as the agent does: /127.5 - 1)
    inp, exp = load_case("agent_raw_image_aloha_b2")
    key = [k for k in inp if k.startswith("obs__") and "image" in k][0]
    frames = f32(inp[key]).reshape(-1, 64, 64, 3) / 127.5 - 1.0
    PV = torch32.TorchParams(vae_params(), dtype=torch.float32)
    PV64 = torch32.TorchParams(vae_params(), dtype=torch.float64)
    ref = torch32.vae_encode_mean(PV64, frames.double()).numpy()
    out["vae_encode_mean_2frames"] = lambda: err(torch32.vae_encode_mean(PV, frames).numpy(), ref)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json")
    ap.add_argument("--only", help="comma-separated arm names")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    cases = run_cases(a.quick)
    arms = [("fp32", None), ("6 acc64", (6, "acc64")), ("6 acc32", (6, "acc32")), ("9 acc32", (9, "acc32")), ("3 acc64", (3, "acc64")),
            ("f16x3 a64", ("f16", "acc64")), ("f16x3 a32", ("f16", "acc32"))]
    if a.quick:
        arms = arms[:3] + arms[5:]
    if a.only:
        arms = [x for x in arms if x[0] in a.only.split(",")]
    res = {}
    real_f = torch32.F
    for arm, cfg in arms:
        torch32.F = real_f if cfg is None else SplitF16(cfg[1]) if cfg[0] == "f16" else SplitF(*cfg)
        for name, fn in cases.items():
            t0 = time.time()
            e = fn()
            res.setdefault(name, {})[arm] = e
            print(f"{name:32s} {arm:9s} max|err| {e:.3e}   ({time.time() - t0:.0f} s)", flush=True)
    torch32.F = real_f
    print()
    print(f"{'golden':32s} " + " ".join(f"{arm:>10s}" for arm, _ in arms))
    for name, r in res.items():
        print(f"{name:32s} " + " ".join(f"{r[arm]:10.2e}" for arm, _ in arms))
    worst6 = max(max(r.get("6 acc64", 0), r.get("6 acc32", 0)) for r in res.values())
    worst16 = max(max(r.get("f16x3 a64", 0), r.get("f16x3 a32", 0)) for r in res.values())
    print(f"worst error of the two-plane fp16 form (three products) over the goldens: {worst16:.2e}  (kill criterion 5e-5: {'PASS' if worst16 <= 5e-5 else 'FAIL'})")
    print(f"\nworst 6-product error over the goldens: {worst6:.2e}  (kill criterion 5e-5: {'PASS' if worst6 <= 5e-5 else 'FAIL'})")
    if a.json:
        json.dump(dict(errors=res, worst_6_product=worst6, worst_f16x3=worst16, kill_criterion=5e-5, passed=worst6 <= 5e-5, passed_f16x3=worst16 <= 5e-5),
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
