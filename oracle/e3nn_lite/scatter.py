"""
``torch_scatter.scatter`` (v2.1.2) semantics for dim=0, reduce in {sum, mean}.
ORACLE / TEST INFRASTRUCTURE.  Reference call sites: nn/conv.py:114, nn/nodewise.py:144.
"""
from typing import Optional

import torch


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None, reduce: str = "sum"):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return out
    if reduce == "mean":
        count = src.new_zeros(dim_size)
        count.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        count = count.clamp(min=1)
        return out / count.reshape((-1,) + (1,) * (src.dim() - 1))
    if reduce in ("min", "max"):   # torch_scatter.scatter_min / scatter_max (values; empty groups stay 0)
        idx = index.reshape((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        init = src.new_full((dim_size,) + tuple(src.shape[1:]), float("inf") if reduce == "min" else float("-inf"))
        red = init.scatter_reduce(0, idx, src, reduce="amin" if reduce == "min" else "amax", include_self=True)
        return torch.where(torch.isinf(red), torch.zeros_like(red), red)
    raise NotImplementedError(reduce)
