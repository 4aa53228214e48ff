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


class DerivedWeight:
    """Caches f(*params) until any parameter changes (optimizer step, load_state_dict, .to())."""

    def __init__(self, fn: Callable[..., torch.Tensor]):
        self._fn = fn
        self._key = None
        self._val = None

    def get(self, *params: torch.Tensor):
        key = tuple((p.data_ptr(), p._version, p.device, p.dtype) for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = self._fn(*params)
            self._key = key
        return self._val
