"""ORACLE (test infrastructure, NOT a product path) -- Philox4x32-10 counter-based generator
(Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
library's `philox4x32_R(10, ctr, key)`), restated in NumPy uint64 arithmetic, plus the Box-Muller
map the HIP path applies to it.

This is the noise source that stands where the reference calls `jax.random.normal`
(agent/ldp_agent.py:461-462,467,475,488-489): JAX's threefry stream is a non-goal,
so the generator is this repo's own and what must be pinned is that it IS Philox4x32-10 --
`KAT` below holds the three known-answer vectors published with Random123 (kat_vectors,
"philox4x32 10" rows) -- and that its normals are N(0,1) (moment / KS tests in tests/test_philox.py).

Keying used by libldp_hip (csrc/tconv.hpp philox_normal, include/ldp_hip.h ldp_philox_*):
    counter = (elem & 0xffffffff, elem >> 32, step, stream_id),  key = (seed & 0xffffffff, seed >> 32)
    planner: elem = ((row_offset + b) * T + t) * 32 + c      IDM: elem = (row_offset + row) * 32 + a, seed ^ 2^63
    stream_id 1 = the initial state x_T / a_T (step 0), stream_id 0 = the scheduler noise of executed step i
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# (counter words, key words) -> output words; Random123 kat_vectors, philox4x32 with 10 rounds
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32-valued array, key: (2,) or (..., 2).  -> (..., 4) uint32."""
    c = np.asarray(ctr, dtype=np.uint64) & MASK
    k = np.broadcast_to(np.asarray(key, dtype=np.uint64) & MASK, c.shape[:-1] + (2,)).copy()
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0], k[..., 1]
    for _ in range(10):
        p0 = M0 * c0                      # 32 x 32 -> 64 bit products fit uint64 exactly
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def words(seed: int, elem0: int, step: int, stream_id: int, n: int) -> np.ndarray:
    """The (n, 4) words libldp_hip's ldp_philox_raw produces for elements elem0 .. elem0+n-1."""
    e = (np.uint64(elem0 & (2**64 - 1)) + np.arange(n, dtype=np.uint64))
    ctr = np.stack([e & MASK, e >> np.uint64(32), np.full(n, step, np.uint64), np.full(n, stream_id, np.uint64)], -1)
    seed &= 2**64 - 1
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))


def normal(seed: int, elem0: int, step: int, stream_id: int, n: int) -> np.ndarray:
    """float64 evaluation of the Box-Muller map of csrc/tconv.hpp (u = (top 24 bits + 0.5) / 2^24)."""
    w = words(seed, elem0, step, stream_id, n).astype(np.float64)
    u1 = (np.floor(w[:, 0] / 256.0) + 0.5) / 16777216.0
    u2 = (np.floor(w[:, 1] / 256.0) + 0.5) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
