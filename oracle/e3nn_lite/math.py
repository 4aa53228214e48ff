"""
``e3nn.math.soft_one_hot_linspace`` (v0.5.1), 'bessel' basis only [e3nn-recalled].
ORACLE / TEST INFRASTRUCTURE.  Reference call site: nn/embedding.py:189-196.
"""
import math

import torch


def soft_one_hot_linspace(x: torch.Tensor, start, end, number, basis=None, cutoff=None):
    if cutoff not in [True, False]:
        raise ValueError("cutoff must be specified")
    if basis != "bessel":
        raise NotImplementedError(f"only basis='bessel' is on the MatTen hot path, got {basis!r}")
    x = x[..., None] - start
    c = end - start
    bessel_roots = torch.arange(1, number + 1, dtype=x.dtype, device=x.device) * math.pi
    out = math.sqrt(2 / c) * torch.sin(bessel_roots * x / c) / x
    if not cutoff:
        return out
    return out * ((x / c) < 1) * (0 < x)
