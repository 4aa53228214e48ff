"""Device-resident copies of immutable plan tables, and version-tracked derived weights."""
from typing import Callable, Dict, Tuple

import numpy as np
import torch


class DeviceTables:
    """numpy tables uploaded lazily, once per device."""

    def __init__(self, **tables: np.ndarray):
        self._host = {k: np.ascontiguousarray(v) for k, v in tables.items()}
        self._dev: Dict[Tuple[str, torch.device], torch.Tensor] = {}

    def get(self, name: str, device) -> torch.Tensor:
        device = torch.device(device)
        key = (name, device)
        t = self._dev.get(key)
        if t is None:
            t = torch.from_numpy(self._host[name]).to(device)
            self._dev[key] = t
        return t


# Writes that torch's version counters cannot see: FlatAdam updates every parameter through the raw pointer of its flat
# buffer, a hipGraph replay re-runs a captured optimiser step, matten_bn_train_fwd updates the running statistics in
# place.  Each of them bumps this process-wide epoch, which is part of every cache key below, so an eval forward after a
# training step never reuses packs derived from the previous weights.
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch() -> None:
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch() -> int:
    return _WEIGHTS_EPOCH[0]


class DerivedWeight:
    """Caches f(*params) until any parameter changes (optimizer step, load_state_dict, .to(), or a raw in-place write
    announced through ``bump_weights_epoch``)."""

    def __init__(self, fn: Callable[..., torch.Tensor]):
        self._fn = fn
        self._key = None
        self._val = None

    def get(self, *params: torch.Tensor):
        key = (_WEIGHTS_EPOCH[0],) + tuple((p.data_ptr(), p._version, p.device, p.dtype) for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = self._fn(*params)
            self._key = key
        return self._val


class WeightSlice:
    """``w.index_select(dim, idx)`` kept in ONE persistent buffer that is refreshed in place whenever the source
    parameter changes (its ``_version`` then moves too, so DerivedWeight caches keyed on the buffer notice)."""

    def __init__(self, idx: np.ndarray, dim: int = 0):
        idx = np.asarray(idx, dtype=np.int64)
        self._idx = DeviceTables(idx=idx)
        self._dim = dim
        self._key = None
        self._buf = None
        self._n = int(idx.size)
        self._identity = bool(idx.size and (idx == np.arange(idx.size)).all())

    def select(self, w: torch.Tensor) -> torch.Tensor:
        """the same slice inside autograd (training): gradients flow back to ``w``, zeros where nothing was selected"""
        if self._identity and w.shape[self._dim] == self._n:
            return w
        return w.index_select(self._dim, self._idx.get("idx", w.device))

    def get(self, w: torch.Tensor) -> torch.Tensor:
        key = (_WEIGHTS_EPOCH[0], w.data_ptr(), w._version, w.device, w.dtype)
        if key != self._key:
            with torch.no_grad():
                new = w.detach().index_select(self._dim, self._idx.get("idx", w.device))
                if self._buf is None or self._buf.device != w.device or self._buf.dtype != w.dtype:
                    self._buf = new.contiguous()
                else:
                    self._buf.copy_(new)
            self._key = key
        return self._buf
