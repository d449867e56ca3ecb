#!/usr/bin/env python3
"""Max abs error of the HIP path against every committed float64 golden (tests/golden/*.npz): how much
of the 1e-4 budget each loop / agent case uses.  Actions are compared in the IDM's normalised space
(where the north-star tolerance is stated) and, for reference, un-normalised.
GPU box only; reads tests/golden, never /root/reference.   usage: parity_margin.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from latent_diffusion_planning_amd.engine import HipEngine
from tests import cfgs
from tests.cases import load_case, unflat_obs, vae_params
from tests.util import idm_params, make_agent, normalised, planner_params


def f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def main():
    out = {}
    for T, cases in ((8, (("planner_loop_ddpm100", "ddpm", 100), ("planner_loop_ddim100", "ddim", 100),
                          ("planner_loop_ddim50", "ddim", 50))), (16, (("planner_loop_t16_ddpm100", "ddpm", 100),))):
        eng = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        eng.load_params(planner=planner_params())
        for name, smp, n in cases:
            inp, exp = load_case(name)
            got = eng.plan_sample(f32(inp["cond"]), x_init=f32(inp["x0"]),
                                  step_noise=f32(inp["nz"]) if smp == "ddpm" else None, sampler=smp, n_steps=n)
            out[name] = {"plan": err(got.cpu().numpy(), exp["plan"])}
        if T == 8:
            inp, exp = load_case("bench_rows_b256_ddim100")
            got = eng.plan_sample(f32(inp["cond"]), x_init=f32(inp["x0"]), sampler="ddim", n_steps=100).cpu().numpy()
            out["bench_rows_b256_ddim100"] = {"plan": err(got[exp["rows"].astype(int)], exp["plan"])}
        eng.close()
    for cfg, D, A in (("rm", 25, 7), ("aloha", 30, 14)):
        pp, ip = planner_params(D=D), idm_params(D=D, A=A)
        eng = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
        eng.load_params(idm=ip)
        for smp, n in (("ddpm", 100), ("ddim", 50)):
            inp, exp = load_case(f"idm_loop_{cfg}_{smp}{n}")
            got = eng.idm_sample(f32(inp["tr"]), a_init=f32(inp["a0"]), step_noise=f32(inp["nz"]) if smp == "ddpm" else None,
                                 sampler=smp, n_steps=n)
            out[f"idm_loop_{cfg}_{smp}{n}"] = {"action_normalised": err(got.cpu().numpy(), exp["act"])}
        eng.close()
        ag, data = make_agent(cfg, pp, ip)
        for B in (1, 5):
            inp, exp = load_case(f"agent_sample_viz_{cfg}_b{B}")
            noise = {k: f32(inp[k]) for k in ("x_init", "x_noise", "a_init", "a_noise")}
            act, met = ag.sample(unflat_obs(inp), 0, noise=noise)
            out[f"agent_sample_viz_{cfg}_b{B}"] = {
                "plan": err(np.array(met["plan"]), exp["plan"]),
                "action_normalised": err(normalised(np.array(act), data), normalised(exp["action"], data)),
                "action_unnormalised": err(np.array(act), exp["action"])}
        inp, exp = load_case(f"agent_training_batch_{cfg}")
        batch = unflat_obs(inp)
        noise = {k: f32(inp[k]) for k in ("x_init", "x_noise", "a_init", "a_noise")}
        act, met = ag.sample(batch, 0, noise=noise)
        sa = ag.sample_action(batch, 0, noise=dict(a_init=f32(inp["a2_init"]), a_noise=f32(inp["a2_noise"])))
        out[f"agent_training_batch_{cfg}"] = {
            "action_normalised": err(normalised(np.array(act), data), normalised(exp["action"], data)),
            "sample_action_normalised": err(normalised(np.array(sa), data), normalised(exp["sample_action"], data)),
            "plan_mse_abs": abs(float(met["plan_mse"]) - float(exp["plan_mse"]))}
        ag._engine.close()
    ag, data = make_agent("rm", planner_params(), idm_params(), T=16)
    inp, exp = load_case("agent_sample_viz_rm_t16_b2")
    act, met = ag.sample(unflat_obs(inp), 0, noise={k: f32(inp[k]) for k in ("x_init", "x_noise", "a_init", "a_noise")})
    out["agent_sample_viz_rm_t16_b2"] = {"plan": err(np.array(met["plan"]), exp["plan"]),
                                         "action_normalised": err(np.array(act), exp["action"])}
    ag._engine.close()
    ag, data = make_agent("rm", planner_params(), idm_params())
    inp, exp = load_case("agent_sample_viz_rm_ddim50_b3")
    act, met = ag.sample(unflat_obs(inp), 0, noise={k: f32(inp[k]) for k in ("x_init", "a_init")}, sampler="ddim", n_steps=50)
    out["agent_sample_viz_rm_ddim50_b3"] = {"plan": err(np.array(met["plan"]), exp["plan"]),
                                            "action_normalised": err(np.array(act), exp["action"])}
    ag._engine.close()
    ag, data = make_agent("aloha", planner_params(D=30), idm_params(D=30, A=14), vae=vae_params())
    inp, exp = load_case("agent_raw_image_aloha_b2")
    batch = unflat_obs(inp)
    act, met = ag.sample(batch, 0, noise={k: f32(inp[k]) for k in ("x_init", "x_noise", "a_init", "a_noise")})
    enc = ag.vae_encode(ag._postprocess(batch)["obs"])["latent_wrist64_image"]
    out["agent_raw_image_aloha_b2"] = {"latent_normalised": err(enc.cpu().numpy(), exp["latent"]),
                                       "plan": err(np.array(met["plan"]), exp["plan"]),
                                       "action_normalised": err(normalised(np.array(act), data), normalised(exp["action"], data)),
                                       "action_unnormalised": err(np.array(act), exp["action"])}
    ag._engine.close()
    worst = max(v for c in out.values() for k, v in c.items() if k != "action_unnormalised")
    doc = {"what": "max |HIP - float64 golden| per case (tests/golden/*.npz; goldens come from this repo's oracle: "
                   "parity unpinned)", "tolerance": 1e-4, "worst_normalised": worst, "cases": out}
    txt = json.dumps(doc, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
