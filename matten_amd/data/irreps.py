"""
Per-module irreps bookkeeping: every backbone module declares the irreps of the dict entries it
reads and writes, and ``Sequential`` checks that consecutive modules agree.  Behavioural mirror
of the reference's ``ModuleIrreps`` mixin (data/irreps.py:17-165); construction-time only.
"""
from typing import Dict, Optional, Sequence

from ..o3 import Irreps
from . import _key

DataKey = _key


def _as_irreps_dict(d: Optional[Dict[str, object]]) -> Dict[str, Optional[Irreps]]:
    return {} if d is None else {k: (None if v is None else Irreps(v)) for k, v in d.items()}


class ModuleIrreps:
    REQUIRED_KEYS_IRREPS_IN: Optional[Sequence[str]] = None

    def init_irreps(self, irreps_in=None, irreps_out=None, *, required_keys_irreps_in: Sequence[str] = None):
        ins = _as_irreps_dict(irreps_in)
        pos = DataKey.POSITIONS
        if ins.get(pos) is not None and ins[pos] != Irreps("1o"):
            raise ValueError(f"Positions must have irreps 1o, got `{ins[pos]}`")
        ins[pos] = Irreps("1o")
        if ins.get(DataKey.EDGE_INDEX) is not None:
            raise ValueError(f"Edge indexes must have irreps `None`, got `{ins[DataKey.EDGE_INDEX]}`")
        ins[DataKey.EDGE_INDEX] = None

        required = list(self.REQUIRED_KEYS_IRREPS_IN or []) + list(required_keys_irreps_in or [])
        for k in required:
            if k not in ins:
                raise ValueError(f"This module {type(self)} requires `{k}` in `irreps_in`.")

        self._irreps_in = ins
        self._irreps_out = dict(ins)
        self._irreps_out.update(_as_irreps_dict(irreps_out))

    @property
    def irreps_in(self):
        return self._irreps_in

    @property
    def irreps_out(self):
        return self._irreps_out


def check_irreps_compatible(a: Dict[str, Irreps], b: Dict[str, Irreps]) -> bool:
    return all(a[k] == b[k] for k in a if k in b)
