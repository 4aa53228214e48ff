"""
Irreps <-> Cartesian conversion of the predicted tensors (mirrors reference utils.py:110-133) and
small config helpers.  ``to_cartesian`` runs on the GPU through the C ABI; ``from_cartesian`` is a
host-side label conversion (dataset preparation, reference dataset/structure_scalar_tensor.py:262-267)
and is plain torch.
"""
from pathlib import Path
from typing import Union

import torch
import yaml

from . import o3, ops


class CartesianTensor(o3.Irreps):
    """Irreps of a symmetric Cartesian tensor, e.g. ``ijkl=jikl=klij`` -> ``2x0e+2x2e+1x4e``."""

    def __new__(cls, formula: str):
        irreps, _ = o3.cartesian_tensor_basis(formula)
        self = tuple.__new__(cls, tuple(irreps))
        self.formula = formula
        self.indices = formula.split("=")[0].replace("-", "")
        return self


class CartesianTensorWrapper:
    def __init__(self, formula: str):
        self.converter = CartesianTensor(formula)
        self.formula = formula
        _, Q = o3.cartesian_tensor_basis(formula)
        self._rank = Q.ndim - 1
        self._Q = torch.from_numpy(Q.reshape(Q.shape[0], -1).copy())  # fp64 [n_irreps_dim, 3^rank]
        self._cache = {}

    def _q(self, dtype, device) -> torch.Tensor:
        key = (dtype, torch.device(device))
        if key not in self._cache:
            self._cache[key] = self._Q.to(dtype=dtype, device=device).contiguous()
        return self._cache[key]

    def from_cartesian(self, data: torch.Tensor) -> torch.Tensor:
        q = self._q(data.dtype, data.device)
        return data.flatten(-self._rank) @ q.T

    def to_cartesian(self, data: torch.Tensor) -> torch.Tensor:
        q = self._q(torch.float32, data.device)
        lead = data.shape[:-1]
        out = ops.dense_rows(data.reshape(-1, data.shape[-1]), q)
        return out.reshape(*lead, *([3] * self._rank))


class ToCartesian(torch.nn.Module):
    def __init__(self, formula: str):
        super().__init__()
        self.ct = CartesianTensorWrapper(formula)

    def forward(self, data: torch.Tensor) -> torch.Tensor:
        return self.ct.to_cartesian(data)


def to_path(path: Union[str, Path]) -> Path:
    return Path(path).expanduser().resolve()


def yaml_load(filename: Union[str, Path]):
    with open(to_path(filename), "r") as f:
        return yaml.safe_load(f)


def yaml_dump(obj, filename: Union[str, Path], sort_keys: bool = False):
    p = to_path(filename)
    p.parent.mkdir(parents=True, exist_ok=True)
    with open(p, "w") as f:
        yaml.dump(obj, f, default_flow_style=False, sort_keys=sort_keys)


def detect_nan_and_inf(x: torch.Tensor, file: Union[str, Path] = None, name: str = None, level: int = 1,
                       filename: str = None) -> None:
    """Raise ValueError if `x` holds a NaN or an Inf (reference utils.py:68-107; the call-site line number the
    reference digs out of the stack is replaced by the caller-supplied `file` / `name`).  One host sync per call."""
    if not (isinstance(x, torch.Tensor) and x.is_floating_point()) or x.numel() == 0:
        return
    bad = "nan" if bool(torch.isnan(x).any()) else "inf" if bool(torch.isinf(x).any()) else None
    if bad is None:
        return
    if filename:
        yaml_dump(x.detach().cpu().numpy().tolist(), filename)
    raise ValueError(f"Tensor is {bad} in {file}, name={name}")
