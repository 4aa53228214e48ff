"""Node-wise output head on MI355X (mirrors reference nn/nodewise.py:18-148)."""
from typing import Dict, Optional

import torch

from .. import ops
from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irreps
from ._nequip import with_batch
from .utils import SpeciesLinear


class NodewiseSelect(ModuleIrreps, torch.nn.Module):
    """Select node features by a boolean mask (reference nn/nodewise.py:18-86): ``data[out_field] =
    data[field][data[mask_field]]``; without a mask field the features are passed through.  Pure row indexing --
    torch's device indexing is the plumbing here, no arithmetic."""

    def __init__(self, irreps_in: Dict[str, Irreps], field: str = DataKey.NODE_FEATURES,
                 out_field: Optional[str] = None, mask_field: Optional[str] = None):
        super().__init__()
        self.field = field
        self.out_field = out_field if out_field is not None else field
        self.mask_field = mask_field
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: irreps_in[self.field]},
                         required_keys_irreps_in=[self.field])

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data = data.copy()  # shallow copy so the input dict is not modified (reference nodewise.py:64-65)
        value = data[self.field]
        data[self.out_field] = value if self.mask_field is None else value[data[self.mask_field]]
        return data

    def __repr__(self):
        return (f"{self.__class__.__name__}(\n  field: {self.field}, out_field: {self.out_field}, out field irreps: "
                f"{self.irreps_out[self.out_field]}\n)")


class NodewiseLinear(ModuleIrreps, torch.nn.Module):
    def __init__(self, irreps_in: Dict[str, Irreps], irreps_out: Irreps = None, field: str = DataKey.NODE_FEATURES,
                 out_field: Optional[str] = None):
        super().__init__()
        self.field = field
        self.out_field = out_field if out_field is not None else field
        if irreps_out is None:
            irreps_out = irreps_in[self.field]
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: irreps_out},
                         required_keys_irreps_in=[self.field])
        self.linear = SpeciesLinear(self.irreps_in[field], None, self.irreps_out[self.out_field])
        self.__dict__["_kept"] = None

    def build_kept_input(self, kept_irreps) -> bool:
        """the same linear for an input row that holds only ``kept_irreps`` (every irrep this linear reads): a second
        plan whose flat weight is an index-selected copy of ``linear.weight`` (see PointConv.build_inference_view)"""
        from .. import plan as _plan
        from ._tables import WeightSlice

        lin = SpeciesLinear(Irreps(kept_irreps), None, self.irreps_out[self.out_field])
        idx = _plan.linear_flat_submap(lin.plan, self.linear.plan)
        if idx is None or idx.size != self.linear.plan.weight_numel:   # it must read every weight the full one reads
            return False
        del lin._parameters["weight"]
        self.__dict__["_kept"] = (lin, WeightSlice(idx))
        return True

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        from .conv import KEPT_ONLY

        if data.get(KEPT_ONLY, False):   # (the marker stays: node_features remain the kept irreps for whoever looks)
            if self._kept is None or self.field != DataKey.NODE_FEATURES:
                raise RuntimeError("node features were pruned for a consumer that has no kept-irreps plan")
            lin, sl = self._kept
            w = self.linear.weight
            grad = torch.is_grad_enabled() and (w.requires_grad or data[self.field].requires_grad)
            lin.__dict__["weight"] = sl.select(w) if grad else sl.get(w)   # (not via Module.__setattr__: select() may return the Parameter)
            try:
                data[self.out_field] = lin(data[self.field])
            finally:
                if grad:
                    lin.__dict__["weight"] = None
            return data
        data[self.out_field] = self.linear(data[self.field])
        return data


class NodewiseReduce(ModuleIrreps, torch.nn.Module):
    def __init__(self, irreps_in: Dict[str, Irreps], field: str, out_field: Optional[str] = None, reduce: str = "sum"):
        super().__init__()
        assert reduce in ("sum", "mean", "min", "max")
        self.reduce = reduce
        self.field = field
        self.out_field = f"{reduce}_{field}" if out_field is None else out_field
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: irreps_in[self.field]},
                         required_keys_irreps_in=[self.field])

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        with_batch(data)
        ptr = data.get(DataKey.PTR)
        if ptr is None:
            # PyG batches are contiguous per crystal: recover segment offsets from `batch`
            batch = data[DataKey.BATCH]
            # the reference scatters by id (nn/nodewise.py:144) and accepts any order; segments need sorted ids
            if batch.numel() > 1 and bool((batch[1:] < batch[:-1]).any()):
                raise ValueError("`batch` must be non-decreasing (nodes grouped per crystal) when `ptr` is not given")
            counts = torch.bincount(batch)
            ptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=batch.device)
            ptr[1:] = torch.cumsum(counts, 0)
        x = data[self.field]
        if self.reduce in ("min", "max"):
            if torch.is_grad_enabled() and x.requires_grad:
                from ..autograd import SegmentMinMaxFn

                data[self.out_field] = SegmentMinMaxFn.apply(x, ptr, self.reduce == "max")
            else:
                data[self.out_field] = ops.segment_minmax(x, ptr, self.reduce == "max")
            return data
        if torch.is_grad_enabled() and x.requires_grad:
            from ..autograd import SegmentReduceFn

            data[self.out_field] = SegmentReduceFn.apply(x, ptr, self.reduce == "mean")
        else:
            data[self.out_field] = ops.segment_reduce(x, ptr, self.reduce == "mean")
        return data
