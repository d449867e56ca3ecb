"""Micro-benchmark helper (run under rocprofv3): repeated launches of single conv primitives so the
weights are L2-warm, to separate MFMA/issue efficiency from HBM/MALL latency effects."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent_diffusion_planning_amd.engine import conv1d_gn_mish_film

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = np.random.default_rng(0)
for (T, cin, cout) in [(2, 1024, 1024), (4, 512, 512), (8, 256, 256), (2, 2048, 512), (4, 1024, 256)]:
    x = torch.tensor(g.standard_normal((B, T, cin)), dtype=torch.float32, device="cuda")
    k = (g.standard_normal((5, cin, cout)) / np.sqrt(5 * cin)).astype(np.float32)
    b = np.zeros(cout, np.float32); s = np.ones(cout, np.float32)
    for _ in range(6):
        y = conv1d_gn_mish_film(x, k, b, s, b)
    torch.cuda.synchronize()
print("done")
