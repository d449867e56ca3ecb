#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations and of the pieces around the planner loop
(IDM loop, VAE encode/decode, end-to-end agent.sample).  Not the driver's bench line (bench.py is)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

if "--lib" in sys.argv:                 # same-box A/B of two builds: --lib PATH loads that libldp_hip
    from latent_diffusion_planning_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from latent_diffusion_planning_amd import flops, weights as W
from latent_diffusion_planning_amd.engine import HipEngine


SPLIT_DTYPE = "f32 (2xfp16 split operands, 3 products, f32 accumulate: x = h + l' / 2^11)"      # what the StableVAE's large 3x3 convs compute in (csrc/sconv.hpp, NPL = 2)
SPLIT6_DTYPE = "f32 (3xbf16 split operands, 6 products, f32 accumulate)"       # option vae_split_f16 = 0: the round's first split form
FP32_DTYPE = "f32 (exact-fp32 MFMA)"
# planner above 256 plans (option planner_split, default on): which layers run on which pipe is part of the label
PLANNER_SPLIT_DTYPE = ("f32 (2xfp16 split operands, 3 exact products, f32 accumulate: x = h + l' / 2^11, range-guarded) for the k=5 / stride-2 / transposed convs of the "
                       "256/512/1024-channel levels above 256 plans; exact-fp32 MFMA for "
                       "the first conv, the 16 -> 8 stride-2 conv of pred_horizon 16 and the final 1x1 conv + step; the IDM's MLPResNet blocks on the same fp16 planes above 256 plans")
PLANNER_BF16_DTYPE = ("f32 (3xbf16 split operands, 6 products, f32 accumulate) for the k=5 / stride-2 / transposed convs of the 256/512/1024-channel levels at T<=8 "
                      "(option planner_split_f16 = 0: the round's first split form); exact-fp32 MFMA for the first conv, T=16 tiles, 1x1 convs and the IDM")


def vae_roofline(flop, dt, split):
    """fp32-equivalent TFLOP/s against the ceiling of the pipe the convs ran on: 2500 / 6 TF/s for six bf16 plane products
    per fp32 multiply-add (`frac` stays <= 1 and comparable), 2500 / 3 for three fp16 plane products (split = "f16"),
    157.3 for the exact-fp32 MFMA."""
    peak = 2500.0 / 3 if split == "f16" else 2500.0 / 6 if split else 157.3
    return dict(bound="mfma-f16x3" if split == "f16" else "mfma-bf16x6" if split else "mfma", achieved=round(flop / dt / 1e12, 2), peak=round(peak, 1), unit="TFLOP/s",
                frac=round(flop / dt / 1e12 / peak, 3), frac_of_fp32_mfma_peak=round(flop / dt / 157.3e12, 3))


def timeit(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    which = sys.argv[1:] or ["idm", "vae", "cfg3", "cfg4", "cfg5", "agent"]
    g = np.random.Generator(np.random.PCG64(0))
    out = {}
    D, A = 25, 7
    pp = W.init_planner_params(W.PlannerSpec(D, D), 0)
    ip = W.init_idm_params(W.IDMSpec(D, A), 1)
    if "idm" in which:
        e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        e.load_params(idm=ip)
        for B in (5, 16, 64, 128, 256, 512, 1024, 2048):
            tr = torch.tensor(g.uniform(-1, 1, (B * 4, 2 * D)), dtype=torch.float32, device="cuda")
            fl = flops.idm_forward_flops(W.IDMSpec(D, A)) * B * 4 * 100
            variants = [("", {}), ("_hs8", {"idm_hs": 8}), ("_hs4", {"idm_hs": 4}), ("_hs2", {"idm_hs": 2}), ("_hs1", {"idm_hs": 1}),
                        ("_unfused", {"idm_unfused": 1}), ("_slicemajor", {"idm_rt_major": 0}), ("_noring", {"idm_noring": 1})]
            if "idm_ablate" in which and B == 256:
                variants += [(f"_dbg{d}", {"dbg": d}) for d in (256, 512, 1024, 2048, 1536, 3840)]
            for tag, opts in variants:
                if tag and B == 5 and tag != "_unfused":
                    continue
                if tag == "_hs1" and B < 256:
                    continue
                for k, v in opts.items():
                    e.set_option(k, v)
                dt = timeit(lambda: e.idm_sample(tr, seed=1), n=5, warm=2)
                _, n_all = e.launch_counts()
                out[f"idm_loop_B{B}{tag}"] = dict(ms=round(dt * 1e3, 2), plans_per_s=round(B / dt, 1), tflops=round(fl / dt / 1e12, 2),
                                                  frac=round(fl / dt / 157.3e12, 3), launches_per_step=round(n_all / 100, 2))
                for k in opts:
                    e.set_option(k, 1 if k == "idm_rt_major" else 0)
        e.close()
    if "idm256" in which:     # only the fused IDM loop at 256 plans (1024 rows): what profiles/r02_kernel_stats_idm_loop_b256.csv traces
        e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        e.load_params(idm=ip)
        tr = torch.tensor(g.uniform(-1, 1, (1024, 2 * D)), dtype=torch.float32, device="cuda")
        dt = timeit(lambda: e.idm_sample(tr, seed=1), n=20, warm=2)
        fl = flops.idm_forward_flops(W.IDMSpec(D, A)) * 1024 * 100
        _, n_all = e.launch_counts()
        out["idm_loop_B256"] = dict(ms=round(dt * 1e3, 2), tflops=round(fl / dt / 1e12, 2), frac=round(fl / dt / 157.3e12, 3),
                                    launches_per_step=round(n_all / 100, 2))
        e.close()
    if "vae" in which:
        vp = W.init_vae_params(seed=2)
        e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        e.load_params(vae=vp)
        for N in (64, 256):
            img = torch.tensor(g.uniform(-1, 1, (N, 64, 64, 3)), dtype=torch.float32, device="cuda")
            dt = timeit(lambda: e.vae_encode(img), n=3, warm=2)
            out[f"vae_encode_N{N}"] = dict(ms=round(dt * 1e3, 2), img_per_s=round(N / dt, 1), dtype=SPLIT_DTYPE,
                                           roofline=vae_roofline(10.988e9 * N, dt, "f16"))
        z = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)), dtype=torch.float32, device="cuda")
        dt = timeit(lambda: e.vae_decode(z), n=3, warm=2)
        out["vae_decode_N64"] = dict(ms=round(dt * 1e3, 2), img_per_s=round(64 / dt, 1), dtype=SPLIT_DTYPE,
                                     roofline=vae_roofline(24.9e9 * 64, dt, "f16"))
        e.close()
    if "small" in which:      # the env-harness regime (eval_bc.yaml: n_eval_processes 5; utils/rm_env_utils.py:150-199 batches 4-5 workers)
        # Bound there: HBM.  One denoising step reads every weight once: planner 262.4 MB + IDM 7.2 MB, whatever B.
        from latent_diffusion_planning_amd.agent import LDPAgent
        from tests import cfgs
        e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        e.load_params(planner=pp, idm=ip)
        data = cfgs.RM_LIFT
        ag = LDPAgent.create(0, None, data["shape_meta"], **cfgs.agent_kwargs(data))
        rows = {}
        for B in (1, 5, 16):
            cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32, device="cuda")
            tp = timeit(lambda: e.plan_sample(cond, seed=1, sampler="ddpm", n_steps=100), n=10, warm=3)
            tr = torch.tensor(g.uniform(-1, 1, (B * 4, 2 * D)), dtype=torch.float32, device="cuda")
            ti = timeit(lambda: e.idm_sample(tr, seed=1), n=10, warm=3)
            batch = cfgs.synth_latent_batch(data, B, 1, 3)
            ta = timeit(lambda: np.array(ag.sample(batch, 1)[0]), n=10, warm=3)
            rows[f"B{B}"] = dict(
                planner_loop_ms=round(tp * 1e3, 2), idm_loop_ms=round(ti * 1e3, 2), agent_sample_ms=round(ta * 1e3, 2),
                plans_per_s=round(B / ta, 1),
                roofline_planner=dict(bound="hbm", bytes_per_step=262.4e6, achieved_GBps=round(262.4e6 * 100 / tp / 1e9, 1),
                                      peak_GBps=6300, peak_spec_GBps=8000, frac=round(262.4e6 * 100 / tp / 6.3e12, 3)),
                roofline_agent=dict(bound="hbm", bytes_per_step=262.4e6 + 7.2e6, achieved_GBps=round((262.4e6 + 7.2e6) * 100 / ta / 1e9, 1),
                                    peak_GBps=6300, peak_spec_GBps=8000, frac=round((262.4e6 + 7.2e6) * 100 / ta / 6.3e12, 3)))
        out["small_batch"] = dict(workload="rm_lift, T = 8, DDPM-100 planner + DDPM-100 IDM, hipGraph, host arrays in / host action out for agent_sample",
                                  note="HBM roofline: every step streams the planner's 262.4 MB (+ IDM 7.2 MB) of fp32 weights once; "
                                       "6.3 TB/s = measured float4 copy rate (MI355X_MICROARCH.md), 8.0 spec", **rows)
        e.close(); ag._engine.close()
    if "vae_ab" in which:     # same-box A/B: StableVAE on split bf16 operands (sconv.hpp) against the exact-fp32 MFMA convs
        vp = W.init_vae_params(seed=2)
        e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        e.load_params(vae=vp)
        img = torch.tensor(g.uniform(-1, 1, (256, 64, 64, 3)), dtype=torch.float32, device="cuda")
        z = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)), dtype=torch.float32, device="cuda")
        for tag, opts in (("fp32", {"vae_split": 0}), ("f16x3", {"vae_split": 1, "vae_split_f16": 1}),
                          ("split6", {"vae_split": 1, "vae_split_f16": 0}),
                          ("split6_resnet_convs_only", {"vae_split": 1, "vae_split_f16": 0, "vae_split_gn_only": 1}),
                          ("f16x3_again", {"vae_split": 1, "vae_split_f16": 1, "vae_split_gn_only": 0}),
                          ("fp32_again", {"vae_split": 0, "vae_split_gn_only": 0})):
            for k, v in opts.items():
                e.set_option(k, v)
            sp = opts.get("vae_split", 1) != 0
            if sp and opts.get("vae_split_f16", 1): sp = "f16"
            dt = timeit(lambda: e.vae_encode(img), n=5, warm=2)
            out[f"vae_encode_N256_{tag}"] = dict(ms=round(dt * 1e3, 2), img_per_s=round(256 / dt, 1), dtype=SPLIT_DTYPE if sp == "f16" else SPLIT6_DTYPE if sp else FP32_DTYPE,
                                                 roofline=vae_roofline(10.988e9 * 256, dt, sp))
            dt = timeit(lambda: e.vae_decode(z), n=5, warm=2)
            out[f"vae_decode_N64_{tag}"] = dict(ms=round(dt * 1e3, 2), img_per_s=round(64 / dt, 1), dtype=SPLIT_DTYPE if sp == "f16" else SPLIT6_DTYPE if sp else FP32_DTYPE,
                                                roofline=vae_roofline(24.9e9 * 64, dt, sp))
        e.set_option("vae_split", 1); e.set_option("vae_split_f16", 1)
        e.close()
    if "cfg3" in which:       # rm_square planner + IDM, T=16, B=1024, DDPM/100, hipGraph
        B = 1024
        obs = torch.tensor(g.uniform(-1, 1, (B, 1, D)), dtype=torch.float32, device="cuda")
        fl = (flops.planner_forward_flops(W.PlannerSpec(D, D), 16) * 100 + flops.idm_forward_flops(W.IDMSpec(D, A)) * 400) * B
        for tag, sp, f16 in (("", 1, 1), ("_planner_bf16x6", 1, 0), ("_planner_fp32", 0, 0)):          # same-box A/B: planner layers on fp16 x 3 (default) | bf16 x 6 | exact fp32
            if tag and "default-only" in which: continue          # (kernel traces of the default engine alone: tools/make_profiles.sh step 5)
            e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=16, action_horizon=4)
            e.set_option("planner_split", sp); e.set_option("planner_split_f16", f16)
            e.load_params(planner=pp, idm=ip)
            dt = timeit(lambda: e.agent_sample(obs, 1, seed=1), n=2)          # ONE captured graph: planner loop -> assembly -> IDM loop
            out["cfg3_T16_B1024_planner+idm" + tag] = dict(ms=round(dt * 1e3, 1), plans_per_s=round(B / dt, 1), tflops=round(fl / dt / 1e12, 2),
                                                           dtype=(PLANNER_SPLIT_DTYPE if f16 else PLANNER_BF16_DTYPE) if sp else FP32_DTYPE,
                                                           roofline=vae_roofline(fl, dt, "f16" if f16 else sp))
            e.close()
    if "cfg5" in which:       # rm_can, 50-step DDIM, 1024 candidates per GPU
        B = 1024
        cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32, device="cuda")
        fl = flops.planner_forward_flops(W.PlannerSpec(D, D), 8) * 50 * B
        for tag, sp, f16 in (("", 1, 1), ("_planner_bf16x6", 1, 0), ("_planner_fp32", 0, 0)):
            e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
            e.set_option("planner_split", sp); e.set_option("planner_split_f16", f16)
            e.load_params(planner=pp)
            dt = timeit(lambda: e.plan_sample(cond, seed=1, sampler="ddim", n_steps=50), n=3)
            out["cfg5_T8_B1024_ddim50" + tag] = dict(ms=round(dt * 1e3, 1), plans_per_s=round(B / dt, 1), tflops=round(fl / dt / 1e12, 2),
                                                     dtype=(PLANNER_SPLIT_DTYPE if f16 else PLANNER_BF16_DTYPE) if sp else FP32_DTYPE,
                                                     roofline=vae_roofline(fl, dt, "f16" if f16 else sp))
            e.close()
    if "cfg4" in which:       # aloha per-GPU shard of configs[3]: raw 64x64 frames -> StableVAE encode -> planner (+ IDM), B = 512
        from latent_diffusion_planning_amd.agent import LDPAgent
        from tests import cfgs
        data = cfgs.ALOHA_CUBE
        ag = LDPAgent.create(0, None, data["shape_meta"], vae_params=W.init_vae_params(seed=2, decoder=False), **cfgs.agent_kwargs(data))
        B = 512
        low = cfgs.synth_latent_batch(data, B, 1, 3)["obs"]
        obs = {k: torch.tensor(v, device="cuda") for k, v in low.items() if not k.startswith("latent_")}
        obs["wrist64_image"] = torch.tensor(g.integers(0, 256, (B, 1, 64, 64, 3)).astype(np.float32), device="cuda")
        batch = {"obs": obs}
        pspec, ispec = W.PlannerSpec(30, 30), W.IDMSpec(30, 14)
        fl = (flops.planner_forward_flops(pspec, 8) * 100 + flops.idm_forward_flops(ispec) * 400 + 10.988e9) * B
        for tag, sp, psp, f16 in (("", 1, 1, 1), ("_all_bf16x6", 1, 1, 0), ("_planner_fp32_vae_bf16x6", 1, 0, 0), ("_all_fp32", 0, 0, 0)):
            ag._engine.set_option("vae_split", sp); ag._engine.set_option("vae_split_f16", f16)
            ag._engine.set_option("planner_split", psp); ag._engine.set_option("planner_split_f16", f16)
            dt = timeit(lambda: ag.sample(batch, 1)[0].tensor, n=3, warm=1)
            out["cfg4_aloha_B512_encode+planner+idm" + tag] = dict(
                ms=round(dt * 1e3, 1), plans_per_s=round(B / dt, 1), tflops=round(fl / dt / 1e12, 2), frac_of_fp32_mfma_peak=round(fl / dt / 157.3e12, 3),
                dtype=("planner " + ((PLANNER_SPLIT_DTYPE if f16 else PLANNER_BF16_DTYPE) if psp else FP32_DTYPE) + "; StableVAE " + ((SPLIT_DTYPE if f16 else SPLIT6_DTYPE) if sp else FP32_DTYPE)),
                note="algorithmic fp32 FLOPs of the whole call against the fp32 MFMA peak (mixed pipes: see dtype)")
        ag._engine.set_option("vae_split", 1); ag._engine.set_option("planner_split", 1); ag._engine.set_option("planner_split_f16", 1); ag._engine.set_option("vae_split_f16", 1)
        ag._engine.close()
    if "agent" in which:      # end-to-end LDPAgent.sample on pre-encoded latents, env-harness batch sizes
        from latent_diffusion_planning_amd.agent import LDPAgent
        from tests import cfgs
        data = cfgs.RM_LIFT
        ag = LDPAgent.create(0, None, data["shape_meta"], **cfgs.agent_kwargs(data))
        for B in (5, 256):
            batch = cfgs.synth_latent_batch(data, B, 1, 3)
            # a policy call as the harness makes it: inputs from the host, action back on the host
            dt = timeit(lambda: np.array(ag.sample(batch, 1)[0]), n=5, warm=2)
            out[f"agent_sample_B{B}"] = dict(ms=round(dt * 1e3, 1), plans_per_s=round(B / dt, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
