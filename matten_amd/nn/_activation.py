"""
Second-moment normalisation constants of the activations (e3nn ``normalize2mom``):
c = (E_{z~N(0,1)} act(z)^2)^(-1/2), estimated exactly like e3nn does -- 10^6 fp64 normals drawn
from ``torch.Generator('cpu').manual_seed(0)`` -- so the constants match to the last digit.
|c-1| < 1e-4 is treated as the identity, as in e3nn.
"""
import functools
import math

import torch
import torch.nn.functional as F

from ..plan import ACT_CODE

_FUNCS = {
    "silu": F.silu,
    "tanh": torch.tanh,
    "sigmoid": torch.sigmoid,
    "ssp": lambda x: F.softplus(x) - math.log(2.0),
    "abs": torch.abs,
}


@functools.lru_cache(maxsize=None)
def normalize2mom_const(name: str) -> float:
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    c = _FUNCS[name](z).pow(2).mean().pow(-0.5).item()
    return 1.0 if abs(c - 1) < 1e-4 else c


def act_const_table() -> torch.Tensor:
    """[8] fp32, indexed by the activation codes of plan.ACT_CODE."""
    t = torch.ones(8, dtype=torch.float32)
    for name, code in ACT_CODE.items():
        if name is not None:
            t[code] = normalize2mom_const(name)
    return t
