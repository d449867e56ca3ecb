#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (float64 torch restatement cross-checked
against oracle/np64.py by tests/test_oracle_kats.py).  Run from the repo root:

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py NAME ...   # selected cases

Each file holds the seeded inputs as float32 (`in_*`, what the HIP path is fed) and the
float64 oracle outputs (`out_*`).  Weights are not stored: they are regenerated from their
seeds by latent_diffusion_planning_amd.weights.init_*_params.
NOTE (parity unpinned): the outputs come from this repository's restatement of the reference
algorithm, not from a run of the JAX reference (not installable here).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.cases import CASES, golden_path  # noqa: E402


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        fn, args = CASES[name]
        inp, compute = fn(*args)
        # the oracle consumes exactly the float32-rounded inputs the HIP path will see
        inp32 = {k: np.asarray(v, dtype=np.float32) for k, v in inp.items()}
        for k in inp:
            inp[k][...] = inp32[k]
        t0 = time.time()
        out = compute()
        np.savez_compressed(golden_path(name), **{f"in_{k}": v for k, v in inp32.items()},
                            **{f"out_{k}": np.asarray(v, dtype=np.float64) for k, v in out.items()})
        print(f"{name}: {time.time() - t0:.1f}s, {os.path.getsize(golden_path(name)) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
