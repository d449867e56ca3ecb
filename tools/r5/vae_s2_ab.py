#!/usr/bin/env python3
"""Round 5 A/B: the StableVAE's stride-2 convs (Downsample2D) and the stride-1 convs of its 8-pixel level on two fp16 planes (option vae_split_s2) against the exact-fp32 tile: encode time and error
against the float64 oracle on the same frames, same box, alternating."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd import weights as W                 # noqa: E402
from latent_diffusion_planning_amd.engine import HipEngine             # noqa: E402
from oracle import torch32                                             # noqa: E402
from tests.util import rng                                             # noqa: E402

vp = W.init_vae_params(seed=2)
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(vae=vp)
g = rng(5)
small = torch.tensor(g.uniform(-1, 1, (4, 64, 64, 3)), dtype=torch.float32)
ref = torch32.vae_encode_mean(torch32.TorchParams(vp, dtype=torch.float64), small.double()).numpy()
img = torch.tensor(g.uniform(-1, 1, (256, 64, 64, 3)), dtype=torch.float32).cuda()
z = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)), dtype=torch.float32).cuda()
zs = z[:3].cpu()
refd = torch32.vae_decode(torch32.TorchParams(vp, dtype=torch.float64), zs.double()).numpy()


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n


for r in range(3):
    for o in (0, 1):
        e.set_option("vae_split_s2", o)
        err = np.abs(e.vae_encode(small.cuda()).cpu().numpy().reshape(ref.shape) - ref).max()
        ms = timeit(lambda: e.vae_encode(img)) * 1e3
        errd = np.abs(e.vae_decode(zs.cuda()).cpu().numpy().reshape(refd.shape) - refd).max()
        msd = timeit(lambda: e.vae_decode(z)) * 1e3
        torch.cuda.synchronize()
        print(f"vae_split_s2={o}: encode N=256 {ms:.3f} ms   max |err| vs float64 (4 frames) {err:.2e}   decode N=64 {msd:.3f} ms  err {errd:.2e}   fault kinds {e.poll_fault_kinds()}", flush=True)
