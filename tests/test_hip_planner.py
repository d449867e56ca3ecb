"""Parity of the HIP planner path (through the C ABI) against the CPU oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import np64, torch32
from tests.util import RM, assert_close, planner_params, rng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(planner=planner_params())
    yield e
    e.close()


@pytest.fixture(scope="module")
def truth():
    return torch32.TorchParams(planner_params(), dtype=torch.float64)


@pytest.mark.parametrize("T,cin,cout,B", [(8, 256, 256, 5), (4, 512, 512, 16), (2, 1024, 1024, 33),
                                          (8, 25, 256, 3), (4, 256, 512, 2), (2, 512, 1024, 2),
                                          (2, 2048, 512, 4), (4, 1024, 256, 4)])
def test_conv_block_primitive(T, cin, cout, B):
    from latent_diffusion_planning_amd.engine import conv1d_gn_mish_film
    g = rng(T * 1000 + cin)
    x = g.standard_normal((B, T, cin))
    p = {"c/Conv_0/kernel": g.standard_normal((5, cin, cout)) / np.sqrt(5 * cin),
         "c/Conv_0/bias": 0.1 * g.standard_normal(cout),
         "c/GroupNorm_0/scale": 1 + 0.1 * g.standard_normal(cout),
         "c/GroupNorm_0/bias": 0.1 * g.standard_normal(cout)}
    film = g.standard_normal((B, 2 * cout))
    ref = np64.conv1d_block(x, p, "c", 8, 5)
    ref_f = film[:, None, :cout] * ref + film[:, None, cout:]
    xt = torch.tensor(x, dtype=torch.float32, device="cuda")
    got = conv1d_gn_mish_film(xt, p["c/Conv_0/kernel"], p["c/Conv_0/bias"], p["c/GroupNorm_0/scale"],
                              p["c/GroupNorm_0/bias"]).cpu().numpy()
    assert_close(got, ref, 1e-5, f"conv+GN+Mish T={T} {cin}->{cout}")
    got = conv1d_gn_mish_film(xt, p["c/Conv_0/kernel"], p["c/Conv_0/bias"], p["c/GroupNorm_0/scale"],
                              p["c/GroupNorm_0/bias"],
                              torch.tensor(film, dtype=torch.float32, device="cuda")).cpu().numpy()
    assert_close(got, ref_f, 2e-5, f"conv+GN+Mish+FiLM T={T} {cin}->{cout}")


@pytest.mark.parametrize("T,c", [(8, 256), (4, 512)])
def test_down_up_primitives(T, c):
    from latent_diffusion_planning_amd.engine import downsample1d, upsample1d
    g = rng(T + c)
    x = g.standard_normal((3, T, c))
    kd, bd = g.standard_normal((3, c, c)) / np.sqrt(3 * c), 0.1 * g.standard_normal(c)
    ref = np64.conv1d(x, kd, bd, 2, np64.same_pads(T, 3, 2))
    got = downsample1d(torch.tensor(x, dtype=torch.float32, device="cuda"), kd, bd).cpu().numpy()
    assert_close(got, ref, 1e-5, f"downsample T={T} C={c}")
    xs = g.standard_normal((3, T // 2, c))
    ku, bu = g.standard_normal((4, c, c)) / np.sqrt(2 * c), 0.1 * g.standard_normal(c)
    ref = np64.conv_transpose1d_same_s2(xs, ku, bu)
    got = upsample1d(torch.tensor(xs, dtype=torch.float32, device="cuda"), ku, bu).cpu().numpy()
    assert_close(got, ref, 1e-5, f"upsample T={T // 2}->{T} C={c}")


@pytest.mark.parametrize("B", [1, 3, 17])
def test_unet_forward_matches_oracle(eng, truth, B):
    g = rng(100 + B)
    x = g.standard_normal((B, 8, 25))
    cond = g.uniform(-1, 1, (B, 25))
    for k in (0, 50, 99):
        ref = torch32.unet_forward(truth, torch.tensor(x), k, torch.tensor(cond)).numpy()
        got = eng.unet_forward(torch.tensor(x, dtype=torch.float32), k, torch.tensor(cond, dtype=torch.float32))
        assert_close(got.cpu().numpy(), ref, 2e-5, f"unet forward B={B} k={k}")
    ks = g.integers(0, 100, size=B)
    ref = torch32.unet_forward(truth, torch.tensor(x), ks, torch.tensor(cond)).numpy()
    got = eng.unet_forward(torch.tensor(x, dtype=torch.float32), torch.tensor(ks), torch.tensor(cond, dtype=torch.float32))
    assert_close(got.cpu().numpy(), ref, 2e-5, f"unet forward B={B} per-sample k")


def test_unet_forward_matches_np64_definition(eng):
    """One evaluation against the explicit-loop float64 definition itself (not the torch one)."""
    g = rng(7)
    x, cond = g.standard_normal((2, 8, 25)), g.uniform(-1, 1, (2, 25))
    ref = np64.unet_forward(np64.to64(planner_params()), x, 37, cond)
    got = eng.unet_forward(torch.tensor(x, dtype=torch.float32), 37, torch.tensor(cond, dtype=torch.float32))
    assert_close(got.cpu().numpy(), ref, 2e-5, "unet forward vs np64")


@pytest.mark.parametrize("sampler,n_steps", [("ddpm", 100), ("ddim", 100), ("ddim", 50)])
def test_plan_sample_matches_oracle(eng, truth, sampler, n_steps):
    """The full loop with explicit noise: north_star tolerance 1e-4 (fp32)."""
    from tests.cases import load_case
    inp, exp = load_case(f"planner_loop_{sampler}{n_steps}")          # tests/golden/*.npz
    cond, x0, nz, ref = inp["cond"], inp["x0"], inp["nz"], exp["plan"]
    for use_graph in (False, True):
        got = eng.plan_sample(torch.tensor(cond, dtype=torch.float32), x_init=torch.tensor(x0, dtype=torch.float32),
                              step_noise=torch.tensor(nz, dtype=torch.float32) if sampler == "ddpm" else None,
                              sampler=sampler, n_steps=n_steps, use_graph=use_graph)
        assert_close(got.cpu().numpy(), ref, 1e-4, f"{sampler}/{n_steps} graph={use_graph}")


def test_graph_replay_is_bit_identical_to_eager(eng):
    g = rng(11)
    cond = torch.tensor(g.uniform(-1, 1, (20, 25)), dtype=torch.float32)
    a = eng.plan_sample(cond, seed=5, sampler="ddpm", use_graph=False)
    b = eng.plan_sample(cond, seed=5, sampler="ddpm", use_graph=True)
    c = eng.plan_sample(cond, seed=5, sampler="ddpm", use_graph=True)     # replay of the cached graph
    d = eng.plan_sample(cond, seed=6, sampler="ddpm", use_graph=True)     # same graph, new seed
    assert torch.equal(a, b) and torch.equal(b, c)
    assert not torch.equal(c, d)
    assert torch.isfinite(d).all() and d.abs().max() <= 1.0 + 1e-4         # clip_sample keeps plans in [-1,1]


def test_philox_stream_is_sharding_invariant(eng):
    """Rows are keyed by their global index: a shard reproduces its slice of the full batch.  Bitwise while full batch and shards run in one
    launch regime (<= 256 | 257..512 | 513..992 | >= 993 plans per call; here 700 plans and two overlapping 560-plan shards); across regimes the summation order changes -- column / K splits below 257 plans, the 32-row tiles from 993 -- and rows
    agree to round-off."""
    g = rng(12)
    cond = torch.tensor(g.uniform(-1, 1, (700, 25)), dtype=torch.float32)
    full = eng.plan_sample(cond, seed=99, sampler="ddpm")                   # 257 .. 992 plans: one regime, one loop
    lo = eng.plan_sample(cond[:560], seed=99, row_offset=0, sampler="ddpm")
    hi = eng.plan_sample(cond[140:], seed=99, row_offset=140, sampler="ddpm")
    assert torch.equal(full[:560], lo) and torch.equal(full[140:], hi)
    for lo_, hi_ in ((24, 32), (100, 172), (300, 500), (200, 600)):        # K-split, quarter groups, half groups (<= 256 plans); 400 plans: the eight-wave tiles of 257..512
        part = eng.plan_sample(cond[lo_:hi_], seed=99, row_offset=lo_, sampler="ddpm")
        eng.check_fault()
        assert_close(part.cpu().numpy(), full[lo_:hi_].cpu().numpy(), 1e-4, f"rows {lo_}:{hi_} as their own batch")


def test_column_split_groups_match_unsplit(eng):
    """B <= 256 splits every GroupNorm group over two work-groups that exchange their partial
    statistics in-launch; B > 256 uses one work-group per group.  Same plans either way (the
    statistics are summed in a different order, so equal to fp32 round-off, not bitwise)."""
    g = rng(13)
    cond = torch.tensor(g.uniform(-1, 1, (272, 25)), dtype=torch.float32)
    unsplit = eng.plan_sample(cond, seed=3, sampler="ddpm")                 # 17 sample blocks
    split = eng.plan_sample(cond[:256], seed=3, sampler="ddpm")             # 16 sample blocks x 2 halves
    eng.check_fault()
    assert_close(split.cpu().numpy(), unsplit[:256].cpu().numpy(), 1e-4, "column split vs unsplit")
    x = torch.tensor(g.standard_normal((272, 8, 25)), dtype=torch.float32)
    e1 = eng.unet_forward(x, 40, cond)
    e2 = eng.unet_forward(x[:200], 40, cond[:200])
    eng.check_fault()
    assert_close(e2.cpu().numpy(), e1[:200].cpu().numpy(), 2e-5, "forward: column split vs unsplit")


@pytest.mark.parametrize("B", [5, 64, 128, 256])
def test_in_launch_exchanges_are_bit_stable_call_after_call(eng, B):
    """The column-split granules (129..256 plans, quarter groups below) and the K-split partial tiles (<= 128 plans)
    are consumed inside the launch that produces them.  Their slabs are reused by every layer of every call, so a
    consumer that ever read a peer's data before it had arrived would see the OTHER seed's values: the same call,
    repeated with alternating seeds, must reproduce its first result bit for bit (summation orders are fixed).
    tools/stress_exchange.py is the long version (1260 hundred-step calls: 0 mismatches on MI355X)."""
    cond = torch.tensor(rng(500 + B).uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
    seeds = (21, 22, 23)
    refs = [eng.plan_sample(cond, seed=s, sampler="ddim", n_steps=50).clone() for s in seeds]
    for i in range(12):
        out = eng.plan_sample(cond, seed=seeds[i % 3], sampler="ddim", n_steps=50)
        assert torch.equal(out, refs[i % 3]), f"call {i}: rows differ from the first result of the same seed"
    eng.check_fault()


_SHARED_GPU = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from latent_diffusion_planning_amd.engine import HipEngine
from tests.util import planner_params
B = int(sys.argv[1])
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params())
for kv in sys.argv[2:]:
    k, v = kv.split("="); e.set_option(k, int(v))
cond = torch.tensor(np.random.default_rng(B).uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
refs = [e.plan_sample(cond, seed=s, sampler="ddim", n_steps=50).clone() for s in (11, 12)]
bad = 0
for i in range(60):
    out = e.plan_sample(cond, seed=11 + i % 2, sampler="ddim", n_steps=50)
    bad += 0 if torch.equal(out, refs[i % 2]) else 1
e.check_fault()
print("SHARED_GPU", B, "mismatches", bad)
"""


@pytest.mark.parametrize("opts", [(), ("safe_mode=1",)])
@pytest.mark.parametrize("sizes", [(256, 200), (16, 64)])
def test_two_processes_sharing_the_gpu_stay_bit_stable(sizes, opts):
    """Two engine processes on one GPU: their launches interleave, work-groups of one launch no longer land on XCD
    (block id % 8), the peers of an in-launch exchange may sit on different XCDs and start far apart.  Every call must
    still reproduce the first result of its seed bit for bit.  Round 2 found this silently broken (10-60 % of the calls
    under sharing, none alone): a poll of the exchange validated one granule while an older poll of the same granule
    was still in flight -- agent-scope loads that miss can be overtaken by later ones that hit, the compiler's partial
    s_waitcnt assumes they cannot -- and the straggler overwrote the validated register with the previous call's value.
    The polls now land completely before anything is looked at; `safe_mode` (no exchange at all) is checked as well."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", _SHARED_GPU, str(b), *opts], cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for b in sizes]
    outs = [p.communicate(timeout=900) for p in procs]
    for (so, se), b in zip(outs, sizes):
        line = [l for l in so.splitlines() if l.startswith("SHARED_GPU")]
        assert line, (so[-1000:], se[-2000:])
        assert line[0] == f"SHARED_GPU {b} mismatches 0", line[0]


def test_k_split_tags_survive_the_epoch_wrap_and_the_slab_wipe():
    """The K-split granules carry 12 bits of the call epoch; the host wipes their slab every 2048 planner calls.
    4200 one-step calls on a fresh handle cross two wipes and the 4096-call wrap of the tag: every call must still
    reproduce the first result of its seed bit for bit."""
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.load_params(planner=planner_params())
    try:
        cond = torch.tensor(rng(77).uniform(-1, 1, (5, 25)), dtype=torch.float32, device="cuda")
        refs = [e.plan_sample(cond, seed=s, sampler="ddim", n_steps=1).clone() for s in (1, 2)]
        bad = 0
        for i in range(4200):
            out = e.plan_sample(cond, seed=1 + i % 2, sampler="ddim", n_steps=1)
            if i % 50 == 0 or 2040 <= i + 2 <= 2060 or 4085 <= i + 2 <= 4110:
                bad += 0 if torch.equal(out, refs[i % 2]) else 1
        e.check_fault()
        assert bad == 0
    finally:
        e.close()


@pytest.mark.parametrize("B", [256, 300, 1043])
def test_xcd_placement_never_changes_a_bit(eng, B):
    """Which XCD a work-group lands on (ConvArgs::by_sample: block index order) is a speed matter only: the same
    tiles are computed in the same order, so plans are bit-identical with the placement rule on (default), off, and
    forced for every layer -- at 256 plans (16 sample blocks), an odd block count and the two-row-block regime."""
    g = rng(15 + B)
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    ref = eng.plan_sample(cond, seed=8, sampler="ddim", n_steps=10)
    try:
        for thr in (0, 1000):
            eng.set_option("by_sample", thr)
            got = eng.plan_sample(cond, seed=8, sampler="ddim", n_steps=10)
            assert torch.equal(got, ref), f"by_sample={thr}"
    finally:
        eng.set_option("by_sample", 2)
    eng.check_fault()


@pytest.mark.parametrize("B", [5, 256, 300])
def test_round3_launch_shapes_agree_with_the_round2_ones(eng, B):
    """Round 3 changed how three layers are LAUNCHED: the final 1x1 conv + scheduler step runs over position pairs
    ((B, T, C) viewed as (B*T/2, 2, C): same rows, same Philox elements) and the transposed convs on half-depth K
    chunks.  Both re-partition K over the waves of a work-group, so the partial sums are added in another order:
    equal to fp32 round-off over the whole 100-step loop, not bitwise -- and each choice is itself bit-stable."""
    g = rng(77 + B)
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    ref = eng.plan_sample(cond, seed=8, sampler="ddpm", n_steps=100)
    assert torch.equal(eng.plan_sample(cond, seed=8, sampler="ddpm", n_steps=100), ref)
    try:
        for opt in ("no_fin_rows", "up_full_depth"):
            eng.set_option(opt, 1)
            got = eng.plan_sample(cond, seed=8, sampler="ddpm", n_steps=100)
            eng.set_option(opt, 0)
            assert_close(got.cpu().numpy(), ref.cpu().numpy(), 2e-5, opt)
    finally:
        eng.set_option("no_fin_rows", 0)
        eng.set_option("up_full_depth", 0)
    eng.check_fault()


def test_two_row_blocks_per_workgroup_are_bit_identical(eng):
    """B >= ~1000 runs the T <= 4 convs with two 16-sample row blocks per work-group (weights fetched
    once for 32 samples).  Per-row arithmetic is unchanged, so rows must equal -- bitwise -- what a
    smaller batch (one row block per work-group, unsplit groups) produces; 1043 = odd number of row
    blocks plus a ragged tail."""
    g = rng(14)
    B = 1043
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    x = torch.tensor(g.standard_normal((B, 8, 25)), dtype=torch.float32)
    eng.set_option("planner_split", 0)               # the exact-fp32 tiles (round 4: above 256 plans the default is split operands, next test)
    eng.set_option("no_batch_split", 1)              # all 1043 plans in ONE loop (1024 + 19 otherwise, engine.hip batch_split)
    try:
        e_big = eng.unet_forward(x, 17, cond)
        for lo, hi in ((0, 272), (771, 1043)):
            e_small = eng.unet_forward(x[lo:hi], 17, cond[lo:hi])
            assert torch.equal(e_big[lo:hi], e_small), f"rows {lo}:{hi}"
        full = eng.plan_sample(cond, seed=21, sampler="ddim", n_steps=10)
        part = eng.plan_sample(cond[768:], seed=21, row_offset=768, sampler="ddim", n_steps=10)   # 275 rows
    finally:
        eng.set_option("no_batch_split", 0)
        eng.set_option("planner_split", 1)
    eng.check_fault()
    assert torch.equal(full[768:], part)


@pytest.mark.parametrize("name,T,smp,n", [("planner_loop_ddpm100", 8, "ddpm", 100), ("planner_loop_ddim50", 8, "ddim", 50),
                                          ("planner_loop_t16_ddpm100", 16, "ddpm", 100)])
@pytest.mark.parametrize("B", [512, 1024])
def test_goldens_tiled_to_shard_sizes_run_the_default_split_regimes(name, T, smp, n, B):
    """The float64 goldens at the batch sizes where the DEFAULT engine takes the split-operand tiles: the golden's rows (conditioning,
    x_T, per-step noise) repeated to 512 plans (16-row tiles + column-split 32-row tiles at T = 2: configs[3]'s shard) and to 1024
    plans (16-row + 32-row tiles: configs[2] / [4]'s shards).  Every copy of a row must match the golden at the unchanged 1e-4."""
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.cases import load_case
    inp, exp = load_case(name)
    b0 = inp["cond"].shape[0]
    idx = np.arange(B) % b0
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)      # noqa: E731
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params())
    assert e.get_option("planner_split") == 1
    got = e.plan_sample(f(inp["cond"][idx]), x_init=f(inp["x0"][idx]), step_noise=f(inp["nz"][:, idx]) if smp == "ddpm" else None,
                        sampler=smp, n_steps=n).cpu().numpy()
    e.check_fault()
    e.set_option("planner_split", 0)
    ref32 = e.plan_sample(f(inp["cond"][idx]), x_init=f(inp["x0"][idx]), step_noise=f(inp["nz"][:, idx]) if smp == "ddpm" else None,
                          sampler=smp, n_steps=n).cpu().numpy()
    e.close()
    assert not np.array_equal(got, ref32), "the split-operand tiles did not run at this batch size"
    assert_close(got, exp["plan"][idx], 1e-4, f"{name} tiled to {B} plans, default engine")
    print(f"{name} x{B}: max|err| split {np.abs(got - exp['plan'][idx]).max():.2e}, exact fp32 {np.abs(ref32 - exp['plan'][idx]).max():.2e}")


def test_split_operand_rows_do_not_depend_on_their_neighbours(eng):
    """Round 4: above 256 plans the k = 5 convs of the 256 / 512 / 1024-channel levels run on split bf16 operands (tconv SPLIT).
    Per-row arithmetic does not depend on the batch inside one launch regime -- 16-row tiles at T = 8 / T = 4 from 257 plans,
    32-row tiles at T = 2 from 993 -- so rows are bit-identical to the same rows as their own sub-batch, whatever work-group and
    row block they land in; across regimes (and against the exact-fp32 path) they agree to round-off."""
    g = rng(15)
    B = 1024
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    x = torch.tensor(g.standard_normal((B, 8, 25)), dtype=torch.float32)
    e_big = eng.unet_forward(x, 17, cond)
    e_tail = eng.unet_forward(x[24:], 17, cond[24:])                 # 1000 rows: same regime, every row in another row block
    assert torch.equal(e_big[24:], e_tail)
    e_mid = eng.unet_forward(x[:900], 17, cond[:900])                # 513..992 plans: 16-row tiles only (T = 2 layers exact fp32)
    for lo, hi in ((0, 600), (290, 900)):
        assert torch.equal(e_mid[lo:hi], eng.unet_forward(x[lo:hi], 17, cond[lo:hi])), f"rows {lo}:{hi}"
    e_low = eng.unet_forward(x[:352], 17, cond[:352])                # 257..352 plans: the eight-wave forms of those tiles (round 5)
    for lo, hi in ((0, 300), (80, 352)):
        assert torch.equal(e_low[lo:hi], eng.unet_forward(x[lo:hi], 17, cond[lo:hi])), f"rows {lo}:{hi}"
    assert_close(e_low.cpu().numpy(), e_mid[:352].cpu().numpy(), 2e-5, "the same rows in the 257..352 and the 513..992 regime")
    eng.set_option("planner_split", 0)
    try:
        e_fp32 = eng.unet_forward(x, 17, cond)
    finally:
        eng.set_option("planner_split", 1)
    assert not torch.equal(e_big, e_fp32), "the split path did not run"
    assert_close(e_big.cpu().numpy(), e_fp32.cpu().numpy(), 2e-5, "split operands against the exact-fp32 kernels, one evaluation")
    assert_close(e_mid.cpu().numpy(), e_fp32[:900].cpu().numpy(), 2e-5, "16-row tiles only against exact fp32")
    assert_close(e_low.cpu().numpy(), e_fp32[:352].cpu().numpy(), 2e-5, "eight-wave 16-row tiles against exact fp32")
    full = eng.plan_sample(cond, seed=21, sampler="ddim", n_steps=10)
    part = eng.plan_sample(cond[24:], seed=21, row_offset=24, sampler="ddim", n_steps=10)
    eng.check_fault()
    assert torch.equal(full[24:], part)
    # around 512 plans (configs[3]'s shard) the T = 2 layers run 32-row split tiles with every GroupNorm group over two
    # work-groups that exchange their partial statistics inside the launch (384 < B <= 512)
    e_512 = eng.unet_forward(x[:512], 17, cond[:512])
    assert torch.equal(e_512[32:], eng.unet_forward(x[32:512], 17, cond[32:512]))          # 480 rows, same regime
    assert not torch.equal(e_512, e_mid[:512]), "the column-split 32-row tiles did not run"
    assert_close(e_512.cpu().numpy(), e_fp32[:512].cpu().numpy(), 2e-5, "512 plans against exact fp32")
    eng.set_option("planner_split_cs2", 0)
    try:
        assert torch.equal(eng.unet_forward(x[:512], 17, cond[:512])[:352], e_low)           # without them: the tiles of 257..352 plans
        eng.set_option("planner_split_8w", 0)
        assert torch.equal(eng.unet_forward(x[:512], 17, cond[:512]), e_mid[:512])          # ... and without the eight-wave forms: those of 513..992
    finally:
        eng.set_option("planner_split_cs2", 1)
        eng.set_option("planner_split_8w", 1)
    eng.check_fault()


@pytest.mark.parametrize("T", [8, 16])
def test_two_row_block_split_tiles_equal_the_one_row_block_tiles(T):
    """From 993 plans the T = 4 layers of 1024 channels (pred_horizon 16) run their 16-row split tiles over TWO row blocks per wave
    (tconv SPLIT = 2, option planner_split_mb2: a weight fragment feeds 32 samples).  Same planes, same K order, same epilogue: plans
    are bit-identical to the one-row-block form's, with a ragged tail (1043 = 32 blocks of 32 + 19: the last work-group's second row
    block holds 3 real samples) and whatever row block a sample lands in.  pred_horizon 8 has no such layer (its T = 4 level has 512
    channels: 64-column groups, measured slower on two row blocks): the option must change nothing there."""
    from latent_diffusion_planning_amd.engine import HipEngine
    g = rng(31 + T)
    B = 1043
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    x = torch.tensor(g.standard_normal((B, T, 25)), dtype=torch.float32)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params())
    e.set_option("no_batch_split", 1)                # all 1043 plans in ONE loop
    assert e.get_option("planner_split_mb2") == 1
    outs, loops, ran = {}, {}, {}
    for v in (1, 0):
        e.set_option("planner_split_mb2", v)
        n0 = e.get_option("stat_mb2_launches")
        outs[v] = e.unet_forward(x, 17, cond)
        ran[v] = e.get_option("stat_mb2_launches") - n0
        loops[v] = e.plan_sample(cond, seed=5, sampler="ddim", n_steps=4)
    e.set_option("planner_split_mb2", 1)
    tail = e.unet_forward(x[16:], 17, cond[16:])     # every sample in the other row block of its wave
    small = e.unet_forward(x[:600], 17, cond[:600])  # below 993 plans: one row block per wave
    e.check_fault()
    e.close()
    assert ran[0] == 0 and (ran[1] > 0) == (T == 16), ran
    assert torch.equal(outs[1], outs[0]), "one evaluation, two row blocks vs one"
    assert torch.equal(loops[1], loops[0]), "DDIM-4 loop, two row blocks vs one"
    assert torch.equal(outs[1][16:], tail)
    if T == 16: assert torch.equal(outs[1][:600], small), "pred_horizon 16 has no 32-row tiles: rows do not depend on the batch regime"


@pytest.mark.parametrize("T", [8, 16])
def test_fp16_planes_agree_with_bf16_planes_and_exact_fp32(T):
    """Above 256 plans the split tiles run on TWO fp16 planes per operand and THREE products by default (tconv SPLIT = 3 / 4: x = h + l' / 2^11,
    option planner_split_f16); the six-product bf16 form stays selectable.  One evaluation at 1043 and 600 plans (both launch regimes, ragged
    tails): each form within 2e-5 of the exact-fp32 kernels, the two forms within 2e-5 of each other; rows of the fp16 form do not depend on
    their neighbours inside a regime."""
    from latent_diffusion_planning_amd.engine import HipEngine
    g = rng(51 + T)
    B = 1043
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32)
    x = torch.tensor(g.standard_normal((B, T, 25)), dtype=torch.float32)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params())
    e.set_option("no_batch_split", 1)
    assert e.get_option("planner_split_f16") == 1
    for n in (B, 600):
        n0 = e.get_option("stat_f16_launches")
        o16 = e.unet_forward(x[:n], 17, cond[:n])
        assert e.get_option("stat_f16_launches") - n0 >= 10, "the fp16 tiles did not run"
        assert torch.equal(o16[40:], e.unet_forward(x[40:n], 17, cond[40:n])) or n - 40 < 993 <= n      # (1043 -> 1003 rows: same regime)
        e.set_option("planner_split_f16", 0)
        n0 = e.get_option("stat_f16_launches")
        ob = e.unet_forward(x[:n], 17, cond[:n])
        assert e.get_option("stat_f16_launches") == n0
        e.set_option("planner_split", 0)
        o32 = e.unet_forward(x[:n], 17, cond[:n])
        e.set_option("planner_split", 1); e.set_option("planner_split_f16", 1)
        assert not torch.equal(o16, ob) and not torch.equal(ob, o32)
        assert_close(o16.cpu().numpy(), o32.cpu().numpy(), 2e-5, f"fp16 x 3 against exact fp32, {n} plans")
        assert_close(ob.cpu().numpy(), o32.cpu().numpy(), 2e-5, f"bf16 x 6 against exact fp32, {n} plans")
        assert_close(o16.cpu().numpy(), ob.cpu().numpy(), 2e-5, f"fp16 x 3 against bf16 x 6, {n} plans")
    e.check_fault()
    e.close()


@pytest.mark.parametrize("name,T,smp,n", [("planner_loop_ddim50", 8, "ddim", 50), ("planner_loop_t16_ddpm100", 16, "ddpm", 100)])
def test_goldens_tiled_to_1024_plans_on_the_bf16_form(name, T, smp, n):
    """The six-product bf16 form (planner_split_f16 = 0) against the float64 goldens at the unchanged 1e-4: it serves the two T = 2 convs with the
    projection by default and everything split on request."""
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.cases import load_case
    inp, exp = load_case(name)
    B = 1024
    idx = np.arange(B) % inp["cond"].shape[0]
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)      # noqa: E731
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params())
    e.set_option("planner_split_f16", 0)
    got = e.plan_sample(f(inp["cond"][idx]), x_init=f(inp["x0"][idx]), step_noise=f(inp["nz"][:, idx]) if smp == "ddpm" else None,
                        sampler=smp, n_steps=n).cpu().numpy()
    e.check_fault()
    e.close()
    assert_close(got, exp["plan"][idx], 1e-4, f"{name} tiled to {B} plans, bf16 form")


# the noise source itself (known-answer vectors, moments, the stream elements the loops draw): tests/test_philox.py


def test_t16_planner_loop_matches_golden():
    """pred_horizon 16 (BASELINE configs[2]): the full 100-step DDPM loop, eager and captured."""
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.cases import load_case
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=16, action_horizon=4)
    e.load_params(planner=planner_params())
    inp, exp = load_case("planner_loop_t16_ddpm100")
    f = lambda a: torch.tensor(a, dtype=torch.float32)     # noqa: E731
    for use_graph in (False, True):
        got = e.plan_sample(f(inp["cond"]), x_init=f(inp["x0"]), step_noise=f(inp["nz"]), use_graph=use_graph)
        assert_close(got.cpu().numpy(), exp["plan"], 1e-4, f"T=16 DDPM-100 graph={use_graph}")
    e.check_fault()
    e.close()


@pytest.mark.parametrize("name,D,A,T,B,sampler,n_steps", [
    ("configs[2] rm_square T=16 planner+IDM", 25, 7, 16, 1024, "ddpm", 100),
    ("configs[3] aloha planner+IDM per-GPU shard", 30, 14, 8, 512, "ddpm", 100),
    ("configs[4] rm_can 50-step DDIM per-GPU shard", 25, 7, 8, 1024, "ddim", 50)])
def test_full_size_configs_size_independent_properties(name, D, A, T, B, sampler, n_steps):
    """BASELINE.json's full per-GPU sizes, through properties that need no oracle run: finite, plans inside
    the clip range, no fault, and rows bit-identical to the same rows sampled as their own (same launch
    regime) sub-batch -- i.e. a plan never depends on its neighbours in the batch."""
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.util import idm_params
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params(D=D), idm=idm_params(D=D, A=A))
    g = rng(B + T + D)
    obs = torch.tensor(g.uniform(-1, 1, (B, 1, D)), dtype=torch.float32)
    x, plan, act = e.agent_sample(obs, 1, seed=17, row_offset=0, sampler=sampler, planner_steps=n_steps,
                                  idm_steps=n_steps)
    e.check_fault()
    assert x.shape == (B, T, D) and plan.shape == (B, 5, D) and act.shape == (B, 4, A)
    for t in (x, plan, act):
        assert torch.isfinite(t).all()
    assert x.abs().max() <= 1.0 + 1e-4 and act.abs().max() <= 1.0 + 1e-4     # clip_sample on the last step
    assert torch.equal(plan[:, 0].cpu(), obs[:, 0]) and torch.equal(plan[:, 1:], x[:, :4])
    # the tail as its own batch, in the same launch regime as the full one (T = 8 model at 1024 plans: its T = 2 layers take the
    # 32-row split tiles from 993 plans, so the tail keeps 1000 rows -- every row in another row block than in the full batch)
    # (512 plans: the T = 2 layers run 32-row split tiles over two work-groups per group from 353 to 512 plans: 480 rows)
    lo = 24 if B >= 1024 else 32      # (a 1000-plan tail: up to 512 plans the T = 4 / T = 8 layers take their eight-wave forms, and 513..767 would run as 512 + rest)
    x2, plan2, act2 = e.agent_sample(obs[lo:], 1, seed=17, row_offset=lo, sampler=sampler, planner_steps=n_steps,
                                     idm_steps=n_steps)
    e.check_fault()
    assert torch.equal(x2, x[lo:])          # (>= 993 plans: two row blocks per work-group, still bit-identical rows)
    # above 256 plans the IDM runs its fp16-plane kernel with four hidden slices whatever the row count (round 5): a row's actions are the
    # same bits in the full batch and in the tail (the exact-fp32 kernel of <= 256 plans slices by row count: round-off there)
    assert torch.equal(act2, act[lo:]), "IDM rows as their own batch"
    e.close()


def test_unsupported_shapes_are_refused_not_miscomputed():
    """down_dims narrower than 256 found a conv instantiation and normalised over the wrong channel
    set; image sizes / tensor shapes were not checked before raw pointers crossed the ABI."""
    from latent_diffusion_planning_amd import weights as W
    from latent_diffusion_planning_amd._lib import LDPHipError
    from latent_diffusion_planning_amd.agent import LDPAgent
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests import cfgs
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4,
                  down_dims=(128, 512, 1024))
    with pytest.raises(LDPHipError, match="down_dims"):
        e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25, down_dims=(128, 512, 1024)), 0))
    with pytest.raises(ValueError, match="img_nhwc"):
        e.vae_encode(torch.zeros(1, 84, 84, 3))
    with pytest.raises(ValueError, match="cond"):
        e.plan_sample(torch.zeros(2, 24))
    with pytest.raises(ValueError, match="transition"):
        e.idm_sample(torch.zeros(4, 49))
    e.close()
    kw = cfgs.agent_kwargs(cfgs.RM_LIFT)
    with pytest.raises(NotImplementedError, match="vae_feature_dim"):          # not one of agent/ldp_agent.py:69-80's four latent shapes
        LDPAgent.create(0, None, cfgs.RM_LIFT["shape_meta"], **dict(kw, vae_feature_dim=48))
    with pytest.raises(NotImplementedError, match="96x96x3 frames"):          # 36 = 3x3x4 latents of 96-pixel frames (built in round 5): 64-pixel frames are refused
        LDPAgent.create(0, None, cfgs.RM_LIFT["shape_meta"], **dict(kw, vae_feature_dim=36))
    with pytest.raises(NotImplementedError, match="128x128x3 frames"):
        LDPAgent.create(0, None, cfgs.RM_LIFT["shape_meta"], **dict(kw, vae_feature_dim=64))
    meta = dict(cfgs.RM_LIFT["shape_meta"], all_shapes=dict(cfgs.RM_LIFT["shape_meta"]["all_shapes"],
                                                            agentview_image=[84, 84, 3]))
    with pytest.raises(NotImplementedError, match="64x64x3 frames"):
        LDPAgent.create(0, None, meta, **kw)
    with pytest.raises(NotImplementedError, match="down_dims"):
        LDPAgent.create(0, None, cfgs.RM_LIFT["shape_meta"], **dict(kw, planner=dict(kw["planner"], down_dims=[128, 256, 512])))


def test_errors_are_reported_not_swallowed(eng):
    from latent_diffusion_planning_amd._lib import LDPHipError
    cond = torch.zeros((2, 25))
    with pytest.raises(LDPHipError, match="n_steps"):
        eng.plan_sample(cond, sampler="ddpm", n_steps=50)
    with pytest.raises(LDPHipError, match="DDIM"):
        eng.plan_sample(cond, sampler="ddim", n_steps=33)
    with pytest.raises(LDPHipError, match="out of range"):
        eng.unet_forward(torch.zeros((2, 8, 25)), 100, cond)


def test_t15_is_rejected_like_the_reference():
    from latent_diffusion_planning_amd._lib import LDPHipError
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=15, action_horizon=4)
    with pytest.raises(LDPHipError, match="multiple of 4"):
        e.load_params(planner=planner_params())
    e.close()


def test_bench_workload_rows_match_golden(eng):
    """The exact bench.py configuration (B=256, 100-step DDIM, column-split grid of 256 work-groups):
    first and last rows of the batch against the float64 oracle (tests/golden/bench_rows_*.npz)."""
    from tests.cases import load_case
    inp, exp = load_case("bench_rows_b256_ddim100")
    got = eng.plan_sample(torch.tensor(inp["cond"], dtype=torch.float32),
                          x_init=torch.tensor(inp["x0"], dtype=torch.float32), sampler="ddim", n_steps=100)
    eng.check_fault()
    rows = exp["rows"].astype(int)
    assert_close(got[rows].cpu().numpy(), exp["plan"], 1e-4, "bench workload rows")


def test_obs_horizon_2_conditioning():
    """global_cond_dim = obs_horizon * D (agent/ldp_agent.py:573-575): FiLM Dense input width 256 + 50."""
    from latent_diffusion_planning_amd.engine import HipEngine
    pp = planner_params(D=25, G=50)
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=50, pred_horizon=8, action_horizon=4)
    e.load_params(planner=pp)
    g = rng(52)
    x, cond = g.standard_normal((3, 8, 25)), g.uniform(-1, 1, (3, 50))
    P = torch32.TorchParams(pp, dtype=torch.float64)
    ref = torch32.unet_forward(P, torch.tensor(x), 12, torch.tensor(cond)).numpy()
    got = e.unet_forward(torch.tensor(x, dtype=torch.float32), 12, torch.tensor(cond, dtype=torch.float32))
    assert_close(got.cpu().numpy(), ref, 2e-5, "unet forward, obs_horizon 2")
    e.close()


def test_empty_and_bad_batches_are_rejected(eng):
    from latent_diffusion_planning_amd._lib import LDPHipError
    with pytest.raises((LDPHipError, ValueError)):
        eng.plan_sample(torch.zeros((0, 25)))
    with pytest.raises(ValueError):
        eng.plan_sample(torch.zeros((2, 25)), x_init=torch.zeros((2, 8, 24)))
    with pytest.raises(ValueError):
        eng.plan_sample(torch.zeros((2, 25)), step_noise=torch.zeros((100, 3, 8, 25)))


@pytest.mark.parametrize("name,T,smp,n", [("planner_loop_ddpm100", 8, "ddpm", 100), ("planner_loop_ddim50", 8, "ddim", 50),
                                          ("planner_loop_t16_ddpm100", 16, "ddpm", 100)])
@pytest.mark.parametrize("ks,cpi", [(1, 2)])          # (round 5: the K-slice / steps-per-stage A/B arms of round 4 are no longer built)
def test_planner_loop_on_split_operands_matches_golden(name, T, smp, n, ks, cpi):
    """Round 4: the k = 5 convs of the 512- / 1024-channel levels on the bf16 matrix pipe with three-plane split operands
    (tconv SPLIT: 32-row tiles on v_mfma_f32_32x32x16_bf16 for T = 2 and plain T = 4, 16-row tiles on v_mfma_f32_16x16x32_bf16
    -- with `ks` K slices and `cpi` / 2 32-channel steps per LDS stage -- for T = 8, T = 4 and the 256-channel level; option
    planner_split, 2 = at any batch, with the column split off so that one work-group owns a GroupNorm group as it does above
    256 plans).  Same goldens, same 1e-4."""
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.cases import load_case
    _f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    inp, exp = load_case(name)
    outs = {}
    for split in (0, 2):
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        e.set_option("planner_split", split)
        e.set_option("no_csplit", 1)
        e.set_option("no_kw", 1)
        e.load_params(planner=planner_params())
        got = e.plan_sample(_f32(inp["cond"]), x_init=_f32(inp["x0"]), step_noise=_f32(inp["nz"]) if smp == "ddpm" else None,
                            sampler=smp, n_steps=n)
        outs[split] = got.cpu().numpy()
        e.check_fault()
        e.close()
    assert not np.array_equal(outs[0], outs[2]), "planner_split did not change the arithmetic: the split path did not run"
    assert_close(outs[2], exp["plan"], 1e-4, f"{name} on split operands")
    # margins on record (profiles/r04_split_planner_margins.json is a copy of this file from the final tree's run)
    import json, os
    path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4", "planner_split_margins.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec[f"{name} ks={ks} cpi={cpi}"] = dict(tolerance=1e-4, max_abs_err_exact_fp32=float(np.abs(outs[0] - exp["plan"]).max()),
                                               max_abs_err_split_operands=float(np.abs(outs[2] - exp["plan"]).max()),
                                               max_abs_split_minus_fp32=float(np.abs(outs[2] - outs[0]).max()))
        json.dump(rec, open(path, "w"), indent=1)
    except OSError:
        pass
    print(f"{name} ks={ks} cpi={cpi}: max|err| exact-fp32 {np.abs(outs[0] - exp['plan']).max():.2e}, split operands {np.abs(outs[2] - exp['plan']).max():.2e}")
