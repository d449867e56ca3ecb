"""DeviceArray -- what `LDPAgent.sample*/vae_*` return.

The reference returns `jnp` arrays and its call sites treat them as numpy-likes:
`np.array(plan_viz)`, `np.array(jax.device_get(batch_action))`, `batch_action[idx]`
(utils/rm_env_utils.py:186-196), `((plan_viz + 1) / 2 * 255).astype(np.uint8).transpose(...)`
(utils/aloha_env_utils.py:96), `jnp.mean(jnp.square(batch['actions'][:, :H] - pred_action))`
(eval_bc.py:135-151), `float(...)`.  A CUDA torch tensor fails every one of those, so results
are wrapped: the data stays in HBM (`.tensor`, what dist.py / bench code use: no host sync) and
is copied to the host on the first numpy-style access, after which it *is* a numpy array for
all practical purposes (`__array__`, arithmetic, indexing, ndarray methods).

Two things ride on the wrapper:
  * laziness -- `metrics['plan_viz']` of `sample()` is a DeviceArray whose tensor is produced by
    a thunk (the VAE decode, 5 x 24.9 GFLOP per plan) only if somebody looks at it
    (eval_bc.py:144 discards it, the video harness consumes it);
  * the fault protocol -- the first host materialisation of any array of a call is that call's
    completion point: it runs the call's `on_complete` hook (poll of the pinned fault word of the
    in-launch exchanges; on a fault the call is recomputed in safe mode and the tensors swapped)
    before any value is handed out.
"""
from __future__ import annotations

import weakref
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch


class CallRecord:
    """One policy call: its arrays and the completion hook shared by all of them."""

    def __init__(self, on_complete: Optional[Callable[["CallRecord"], None]] = None):
        self._refs: List["weakref.ref"] = []      # weak: array -> record is the only strong edge (no reference cycles)
        self.on_complete = on_complete
        self.completed = False
        self._running = False         # the hook is executing (its own tensor reads must not re-enter it)
        self.seqs = ()                # per engine of the agent: its call sequence number right after this call was enqueued

    @property
    def arrays(self) -> List[Optional["DeviceArray"]]:
        """The call's arrays in creation order (None for one that is already gone)."""
        return [r() for r in self._refs]

    def complete(self):
        """Runs the completion hook once.  The record counts as completed only after the hook returned: if the
        safe-mode recompute raises, the arrays keep refusing to hand out the faulted tensors (the next host access
        runs the hook again) instead of silently serving them."""
        if self.completed or self._running:
            return
        hook = self.on_complete
        if hook is not None:
            self._running = True
            try:
                hook(self)
            finally:
                self._running = False
            self.on_complete = None
        self.completed = True


class DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    __array_priority__ = 100

    def __init__(self, tensor: Optional[torch.Tensor] = None, *, thunk: Optional[Callable[[], torch.Tensor]] = None,
                 shape: Optional[Tuple[int, ...]] = None, record: Optional[CallRecord] = None):
        assert (tensor is None) != (thunk is None)
        self._tensor = tensor
        self._thunk = thunk
        self._shape = tuple(tensor.shape) if tensor is not None else tuple(shape)
        self._host: Optional[np.ndarray] = None
        self._record = record
        if record is not None:
            record._refs.append(weakref.ref(self))

    # -- device side ---------------------------------------------------------------------------------
    @property
    def tensor(self) -> torch.Tensor:
        """The device tensor (runs the producing thunk first if this array is lazy).  No host sync."""
        if self._tensor is None:
            self._tensor = self._thunk()
        return self._tensor

    def _swap(self, tensor: Optional[torch.Tensor]):
        """Replace the payload (fault recovery).  A lazy array forgets what it produced and re-runs its
        thunk (which reads the swapped inputs) on the next access."""
        self._host = None
        if tensor is not None:
            self._tensor = tensor
        elif self._thunk is not None:
            self._tensor = None

    def complete(self) -> "DeviceArray":
        """Completion point without a host copy: waits for the producing stream and runs the call's completion hook
        (fault poll; safe-mode recompute and tensor swap on a fault).  For consumers that keep the data on the device
        but let it leave the call -- dist.sample_sharded before its all-gather."""
        if self._record is not None and not self._record.completed:
            t = self._tensor                                   # a lazy array that was never produced has nothing in flight
            if t is not None and t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
            self._record.complete()
        return self

    def cpu(self) -> torch.Tensor:
        return torch.from_numpy(self.numpy())

    # -- host side -----------------------------------------------------------------------------------
    def numpy(self) -> np.ndarray:
        if self._host is None:
            t = self.tensor
            host = t.detach().to("cpu").numpy()                # synchronises with the producing stream
            if self._record is not None and not self._record.completed:
                self._record.complete()                        # may swap self._tensor (fault recovery)
                if self._tensor is not t:
                    host = self.tensor.detach().to("cpu").numpy()
            self._host = host
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and a.dtype != dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        conv = lambda x: x.numpy() if isinstance(x, DeviceArray) else x     # noqa: E731
        if "out" in kwargs:
            kwargs["out"] = tuple(conv(o) for o in kwargs["out"])
        return getattr(ufunc, method)(*[conv(x) for x in inputs], **kwargs)

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def size(self):
        return int(np.prod(self._shape, dtype=np.int64))

    @property
    def dtype(self):
        return np.dtype(np.float32)

    def __len__(self):
        return self._shape[0]

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __iter__(self):
        return iter(self.numpy())

    def __float__(self):
        return float(self.numpy())

    def __repr__(self):
        state = "host" if self._host is not None else ("lazy" if self._tensor is None else "device")
        return f"DeviceArray(shape={self._shape}, {state})"

    def __getattr__(self, name):
        # ndarray methods / attributes the wrapper does not define (astype, transpose, reshape, mean, T, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)


def as_tensor(x, device=None) -> torch.Tensor:
    """DeviceArray / tensor / array-like -> torch tensor (device tensor untouched for a DeviceArray)."""
    if isinstance(x, DeviceArray):
        t = x.tensor
    elif torch.is_tensor(x):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x, dtype=np.float32))
    return t if device is None else t.to(device)
