"""``Sequential`` over dict-passing modules with an irreps hand-off check (reference nn/sequential.py:9-48)."""
from collections import OrderedDict

import torch

from ..data.irreps import ModuleIrreps, check_irreps_compatible


class Sequential(torch.nn.Sequential, ModuleIrreps):
    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            named = args[0]
        else:
            named = OrderedDict((f"{m.__class__.__name__}_{i}", m) for i, m in enumerate(args))
        mods = list(named.values())
        for i in range(len(mods) - 1):
            if not check_irreps_compatible(mods[i].irreps_out, mods[i + 1].irreps_in):
                raise ValueError(
                    f"Output irreps of module {i} `{type(mods[i]).__name__}`: {mods[i].irreps_out}` is incompatible "
                    f"with input irreps of module {i + 1} `{type(mods[i + 1]).__name__}`: {mods[i + 1].irreps_in}."
                )
        self.init_irreps(irreps_in=mods[0].irreps_in, irreps_out=mods[-1].irreps_out)
        torch.nn.Sequential.__init__(self, named)
