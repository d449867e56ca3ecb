"""Host logic of the fault protocol (agent.py `_record` / `_guarded`) on stand-in engines: which calls are recomputed, on which
engines' faults, with which warning -- no GPU.  The device side is covered by tests/test_hip_stress.py and test_hip_idm_agent.py."""
import warnings

import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd.agent import LDPAgent
from latent_diffusion_planning_amd.arrays import DeviceArray
from latent_diffusion_planning_amd.engine import HipEngine
from latent_diffusion_planning_amd._lib import LDPHipFault


class FakeEngine:
    FAULT_EXCHANGE, FAULT_RANGE = HipEngine.FAULT_EXCHANGE, HipEngine.FAULT_RANGE

    def __init__(self):
        self.call_seq, self.fault_upto, self.fault_kinds, self.last_fault_kinds = 0, -1, 0, 0
        self.pending = 0              # what the pinned words hold
        self.refuse = 0               # entry_fault_check: unacknowledged kinds

    def poll_fault_kinds(self):
        f, self.pending, self.refuse = self.pending | self.refuse, 0, 0
        if f:
            self.fault_upto = self.call_seq
            self.fault_kinds |= f
            self.last_fault_kinds = f
        return f

    def enqueue(self):
        if self.pending or self.refuse:
            self.refuse |= self.pending
            self.pending = 0
            raise LDPHipFault(-6, "pending")
        self.call_seq += 1


class Agent(LDPAgent):
    def __init__(self, engines):
        self._engs = engines
        self._device = torch.device("cpu")

    def _engines(self):
        return self._engs


@pytest.fixture(autouse=True)
def _no_cuda_sync(monkeypatch):
    class S:
        def synchronize(self):
            pass
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: S())


def _call(ag, value):
    def run():
        for e in ag._engs:
            e.enqueue()
        return [torch.full((2,), float(value))]
    rec = ag._record(run)
    res = ag._guarded(run)
    rec.seqs = ag._seqs()
    return DeviceArray(res[0], record=rec), run


def test_a_clean_call_is_not_recomputed():
    e = FakeEngine()
    ag = Agent([e])
    arr, _ = _call(ag, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert np.array_equal(np.array(arr), [1, 1])
    assert e.call_seq == 1


@pytest.mark.parametrize("kind,msg", [(HipEngine.FAULT_EXCHANGE, "safe mode"), (HipEngine.FAULT_RANGE, "three bf16 planes")])
def test_a_fault_on_either_engine_recomputes_the_call(kind, msg):
    """LDPHierAgent has two handles; a fault on the SECOND used to go unnoticed (only the first was polled)."""
    e1, e2 = FakeEngine(), FakeEngine()
    ag = Agent([e1, e2])
    arr, _ = _call(ag, 1)
    other, _ = _call(ag, 2)                      # enqueued before the fault is found: condemned as well
    e2.pending = kind
    with pytest.warns(RuntimeWarning, match=msg):
        np.array(arr)
    assert e1.call_seq == 3 and e2.call_seq == 3, "both engines' calls were re-issued once"
    with pytest.warns(RuntimeWarning):
        np.array(other)
    assert e2.call_seq == 4
    later, _ = _call(ag, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        np.array(later)


def test_a_recompute_that_meets_the_other_kind_of_fault_is_repeated():
    e = FakeEngine()
    ag = Agent([e])
    n = {"runs": 0}

    def run():
        e.enqueue()
        n["runs"] += 1
        if n["runs"] == 2:                        # the first recompute (safe mode) trips the range guard
            e.pending = HipEngine.FAULT_RANGE
        return [torch.zeros(1)]
    rec = ag._record(run)
    res = ag._guarded(run)
    rec.seqs = ag._seqs()
    arr = DeviceArray(res[0], record=rec)
    e.pending = HipEngine.FAULT_EXCHANGE
    with pytest.warns(RuntimeWarning):
        np.array(arr)
    assert n["runs"] == 3 and e.fault_kinds == 3


def test_an_unread_faulted_call_does_not_wedge_the_second_engine():
    """_guarded acknowledges on EVERY engine: a refusing second handle used to raise four times and stay refused."""
    e1, e2 = FakeEngine(), FakeEngine()
    ag = Agent([e1, e2])
    first, _ = _call(ag, 1)
    e2.pending = HipEngine.FAULT_EXCHANGE
    second, _ = _call(ag, 2)                      # e2 refuses once, is acknowledged, the call goes through
    assert e2.fault_upto >= 1
    with pytest.warns(RuntimeWarning):
        np.array(first)
    np.array(second)


def test_a_warning_names_the_fault_that_condemned_the_call_not_every_fault_ever_seen():
    """the warning was chosen from the engine's STICKY fault mask, so after one range fault every later exchange fault printed both."""
    e = FakeEngine()
    ag = Agent([e])
    first, _ = _call(ag, 1)
    e.pending = HipEngine.FAULT_RANGE
    with pytest.warns(RuntimeWarning, match="three bf16 planes"):
        np.array(first)
    second, _ = _call(ag, 2)
    e.pending = HipEngine.FAULT_EXCHANGE
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        np.array(second)
    texts = [str(w.message) for w in seen]
    assert len(texts) == 1 and "safe mode" in texts[0] and "bf16" not in texts[0], texts
    assert e.fault_kinds == HipEngine.FAULT_RANGE | HipEngine.FAULT_EXCHANGE
