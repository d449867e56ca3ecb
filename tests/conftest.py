import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """On a GPU box: leave the list of tconv instantiations this pytest process launched in gpurun_out/plans_used_by_suite.txt
    (the library keeps a process-wide log, option dump_plans = 2): the evidence behind the trim of the instantiation lists
    (profiles/r05_plans_used.txt).  Never fails the run."""
    if not _have_gpu():
        return
    try:
        import contextlib
        import io
        import tempfile
        from latent_diffusion_planning_amd.engine import HipEngine
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
        # the library prints to the C stderr: redirect fd 2 into a file for the duration of the dump
        with tempfile.TemporaryFile(mode="w+b") as tmp:
            sys.stderr.flush()
            saved = os.dup(2)
            os.dup2(tmp.fileno(), 2)
            try:
                e.set_option("dump_plans", 2)
            finally:
                os.dup2(saved, 2)
                os.close(saved)
            tmp.seek(0)
            lines = sorted(set(ln.split(" launches=")[0] for ln in tmp.read().decode().splitlines() if ln.startswith("PLAN ")))
        e.close()
        with open(os.path.join(out, "plans_used_by_suite.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except Exception:                                                  # noqa: BLE001
        pass
