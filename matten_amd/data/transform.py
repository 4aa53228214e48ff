"""
Target standardisation for tensor-valued labels (the `normalizer` a task hands to
``ScalarTensorModel.transform_prediction``).  Behavioural mirror of the reference's data/transform.py:59-287:

* ``MeanNormNormalize(irreps)``: per irrep copy, like e3nn's BatchNorm -- scalar (0e) channels are centred and
  divided by their standard deviation, every other irrep is divided by the root of its mean squared component
  (``normalization="component"``) or squared norm (``"norm"``), reduced over the samples by ``mean`` or ``max``.
* ``ScalarNormalize(num_features)``: per-feature mean / population standard deviation (a constant feature keeps
  scale 1, as scikit-learn's StandardScaler does in the reference).

Both keep ``mean`` and ``norm`` as buffers of the full feature width, so a state_dict written by the reference
loads unchanged; ``forward`` = (x - mean) / (norm * scale), ``inverse`` undoes it.  Elementwise host/device tensor
math on [B, D] labels: no kernel of the hot path is involved.
"""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from ..o3 import Irreps


class _Standardizer(torch.nn.Module):
    def __init__(self, width: int, mean: Optional[Tensor], norm: Optional[Tensor], scale: float):
        super().__init__()
        self.scale = scale
        self.mean_norm_initialized = mean is not None and norm is not None
        self.register_buffer("mean", torch.zeros(width) if mean is None else mean)
        self.register_buffer("norm", torch.zeros(width) if norm is None else norm)

    def _require_statistics(self):
        if not self.mean_norm_initialized:
            raise RuntimeError("mean and norm not initialized.")

    def forward(self, data: Tensor) -> Tensor:
        self._require_statistics()
        return (data - self.mean) / (self.norm * self.scale)

    def inverse(self, data: Tensor) -> Tensor:
        self._require_statistics()
        return data * (self.norm * self.scale) + self.mean

    def load_state_dict(self, state_dict, strict: bool = True):
        out = super().load_state_dict(state_dict, strict)
        self.mean_norm_initialized = True
        return out


class MeanNormNormalize(_Standardizer):
    def __init__(self, irreps: Union[str, Irreps], mean: Tensor = None, norm: Tensor = None,
                 normalization: str = "component", reduce: str = "mean", eps: float = 1e-5, scale: float = 1.0):
        self.irreps = Irreps(irreps)
        if normalization not in ("component", "norm"):
            raise ValueError(f"Invalid normalization option {normalization}")
        if reduce not in ("mean", "max"):
            raise ValueError(f"Invalid reduce option {reduce}")
        super().__init__(self.irreps.dim, mean, norm, scale)
        self.normalization, self.reduce, self.eps = normalization, reduce, eps

    def compute_statistics(self, data: Tensor) -> Tuple[Tensor, Tensor]:
        """mean [D] and norm [D] of `data` [B, D] (D = irreps.dim), stored as the module's statistics."""
        assert data.shape[-1] == self.irreps.dim, (data.shape, self.irreps.dim)
        means, norms, col = [], [], 0
        for mul, ir in self.irreps:
            d = ir.dim
            block = data[:, col : col + mul * d].reshape(-1, mul, d)
            col += mul * d
            if ir.is_scalar():
                mu = block.mean(dim=0).reshape(mul)
                block = block - mu.reshape(1, mul, 1)
            else:
                mu = torch.zeros(mul, dtype=data.dtype)
            sq = block.pow(2)
            per_sample = sq.mean(dim=-1) if self.normalization == "component" else sq.sum(dim=-1)  # [B, mul]
            reduced = per_sample.mean(dim=0) if self.reduce == "mean" else per_sample.max(dim=0).values
            means.append(mu.repeat_interleave(d))
            norms.append((reduced + self.eps).sqrt().repeat_interleave(d))
        mean, norm = torch.cat(means), torch.cat(norms)
        self.load_state_dict({"mean": mean, "norm": norm})
        return mean, norm


class ScalarNormalize(_Standardizer):
    def __init__(self, num_features: int, mean: Tensor = None, norm: Tensor = None, scale: float = 1.0):
        super().__init__(num_features, mean, norm, scale)

    def compute_statistics(self, data: Tensor) -> Tuple[Tensor, Tensor]:
        assert data.ndim == 2, "Can only deal with tensor [N_samples, N_features]"
        mean = data.mean(dim=0)
        std = data.std(dim=0, unbiased=False)
        std = torch.where(std == 0, torch.ones_like(std), std)  # StandardScaler leaves constant features unscaled
        self.load_state_dict({"mean": mean, "norm": std})
        return mean, std


class _DelayedStatistics(torch.nn.Module):
    """Shared part of the two target transforms below: the statistics file is read the first time `inverse` (or
    `forward`) runs, like the reference (data/transform.py:477-488, 577-589), unless the statistics were handed over
    directly (a checkpoint that carries the filled normalizer)."""

    def __init__(self, dataset_statistics_path=None):
        super().__init__()
        self.dataset_statistics_path = dataset_statistics_path
        self.dataset_statistics_loaded = False

    def _statistics(self):
        if self.dataset_statistics_path is None:
            raise ValueError("Cannot load dataset statistics from file `None`")
        return torch.load(self.dataset_statistics_path, map_location="cpu", weights_only=True)


class TensorTargetTransform(_DelayedStatistics):
    """reference data/transform.py:520-590: MeanNormNormalize of ONE tensor target given in irreps space"""

    def __init__(self, target_name: str = "elastic_tensor_full", dataset_statistics_path=None, scale: float = 1.0,
                 irreps: str = "2x0e+2x2e+4e"):
        super().__init__(dataset_statistics_path)
        self.target_name = target_name
        self.normalizer = MeanNormNormalize(irreps=irreps, scale=scale)

    def _fill_state_dict(self, device):
        if not self.dataset_statistics_loaded:
            if not self.normalizer.mean_norm_initialized:
                self.normalizer.load_state_dict(self._statistics()[self.target_name])
            self.dataset_statistics_loaded = True
        if self.normalizer.mean.device != torch.device(device):
            self.to(device)

    def forward(self, target: Tensor) -> Tensor:
        """the reference takes a Crystal and rewrites struct.y[target_name]; the dataset layer is out of scope here, so
        this is the same map on the bare [.., D] target"""
        self._fill_state_dict(target.device)
        return self.normalizer(target)

    def inverse(self, data: Tensor) -> Tensor:
        self._fill_state_dict(data.device)
        return self.normalizer.inverse(data)


class ScalarTargetTransform(_DelayedStatistics):
    """reference data/transform.py:418-517: one ScalarNormalize(1) per scalar target name"""

    def __init__(self, target_names, dataset_statistics_path=None):
        super().__init__(dataset_statistics_path)
        self.target_names = list(target_names)
        self.normalizers = torch.nn.ModuleDict({name: ScalarNormalize(num_features=1) for name in self.target_names})

    def _fill_state_dict(self, device):
        if not self.dataset_statistics_loaded:
            stats = None
            for name in self.target_names:
                if not self.normalizers[name].mean_norm_initialized:
                    stats = stats if stats is not None else self._statistics()
                    self.normalizers[name].load_state_dict(stats[name])
            self.dataset_statistics_loaded = True
        if any(n.mean.device != torch.device(device) for n in self.normalizers.values()):
            self.to(device)

    def forward(self, target: Tensor, target_name: str) -> Tensor:
        self._fill_state_dict(target.device)
        return self.normalizers[target_name](target)

    def inverse(self, data: Tensor, target_name: str) -> Tensor:
        self._fill_state_dict(data.device)
        return self.normalizers[target_name].inverse(data)
