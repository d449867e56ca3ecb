"""The committed golden fixtures are reproducible from the oracle (CPU): re-evaluates the cheap
cases bit-for-bit and checks every fixture's stored inputs against its seed."""
import numpy as np
import pytest

from oracle import np64
from tests.cases import CASES, load_case
from tests.util import idm_params


@pytest.mark.parametrize("name", sorted(CASES))
def test_fixture_loads_and_inputs_match_seed(name):
    inp, exp = load_case(name)
    assert exp and all(np.isfinite(v).all() for v in exp.values())


@pytest.mark.parametrize("name", ["idm_loop_rm_ddpm100", "idm_loop_aloha_ddim50", "planner_loop_ddim50"])
def test_oracle_reproduces_fixture(name):
    fn, args = CASES[name]
    inp, compute = fn(*args)
    for k in inp:                                   # the fixture was generated from float32-rounded inputs
        inp[k][...] = np.asarray(inp[k], dtype=np.float32)
    out = compute()
    _, exp = load_case(name)
    for k, v in out.items():
        np.testing.assert_allclose(v, exp[k], rtol=0, atol=1e-12)


def test_fixture_agrees_with_the_numpy_definition():
    """idm_loop fixtures come from the torch-float64 restatement; the explicit NumPy loop
    (oracle/np64.py) must land on the same numbers."""
    inp, exp = load_case("idm_loop_rm_ddpm100")
    ref = np64.idm_sample(idm_params(), np.asarray(inp["tr"], np.float32), np.asarray(inp["a0"], np.float32),
                          np.asarray(inp["nz"], np.float32), 100, 100, "ddpm")
    np.testing.assert_allclose(ref, exp["act"], rtol=0, atol=5e-6)
