"""Chain dict-passing modules, threading irreps_out -> irreps_in (mirrors reference model_factory/utils.py:13-91)."""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

from ..nn.sequential import Sequential
from ..o3 import Irreps


def create_sequential_module(
    modules: "OrderedDict[str, Tuple[type, Dict]]",
    irreps_in: Optional[Dict[str, Irreps]] = None,
    use_kwargs_irreps_in: bool = False,
) -> Sequential:
    built = OrderedDict()
    prev = None
    for name, (cls_type, kwargs) in modules.items():
        ir = irreps_in if prev is None else prev.irreps_out
        if "irreps_in" in kwargs:
            if not use_kwargs_irreps_in:
                raise ValueError(
                    f"Trying to automatically determine irreps_in for module {name} "
                    f"But it is provided as kwargs. Set `use_kwargs_irrpes_in=True` to force it."
                )
            ir = dict(ir or {})
            ir.update(kwargs["irreps_in"])
        kw = dict(kwargs)
        kw["irreps_in"] = ir
        try:
            prev = cls_type(**kw)
        except Exception as e:
            raise RuntimeError(f"Failed instantiate module `{cls_type.__name__}` with kwargs: `{kw}`") from e
        built[name] = prev
    return Sequential(built)
