"""
Edge geometry for the backbone (mirrors the NequIP-derived helpers of reference nn/_nequip.py).

  with_edge_vectors / SphericalHarmonicEdgeAttrs  reference nn/_nequip.py:130-176, 214-268
  with_batch                                      reference nn/_nequip.py:272-285

On MI355X one fused kernel produces, per edge in destination-sorted order, the displacement
vector, its length and the real spherical harmonics; the dst-sorted CSR (built once per batch)
and those arrays are stashed in the data dict for every conv layer to reuse.  The reference's
``edge_vectors`` / ``edge_attrs`` / ``edge_embedding`` entries are only materialised on request.
"""
from typing import Union

import torch

from .. import ops
from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irreps


def with_batch(data: DataKey.Type) -> DataKey.Type:
    if DataKey.BATCH not in data:
        pos = data[DataKey.POSITIONS]
        data[DataKey.BATCH] = torch.zeros(len(pos), dtype=torch.long, device=pos.device)
    return data


def ensure_graph(data: DataKey.Type, err=None) -> DataKey.Type:
    """Destination-sorted CSR of edge_index, built once per batch (err: optional zeroed int32[1] for its range flag)."""
    if DataKey.AMD_ROWPTR not in data:
        n_nodes = data[DataKey.POSITIONS].shape[0]
        perm, rowptr, src, err = ops.csr_build(data[DataKey.EDGE_INDEX], n_nodes, err=err)
        data[DataKey.AMD_PERM], data[DataKey.AMD_ROWPTR], data[DataKey.AMD_SRC] = perm, rowptr, src
        data["_amd_csr_err"] = err
    return data


def ensure_edge_geometry(data: DataKey.Type, lmax: int = None, want_vectors=False, want_lengths=False,
                         want_attrs=False, want_embedding=False) -> DataKey.Type:
    have = DataKey.AMD_GEOM in data
    missing = (
        (want_vectors and DataKey.EDGE_VECTORS not in data)
        or (want_lengths and DataKey.EDGE_LENGTH not in data)
        or (want_attrs and DataKey.EDGE_ATTRS not in data)
        or (want_embedding and DataKey.EDGE_EMBEDDING not in data)
    )
    if have and not missing:
        return data
    ensure_graph(data)
    if lmax is None:
        lmax = int(data["_amd_lmax"]) if "_amd_lmax" in data else 0
    data["_amd_lmax"] = lmax
    nb, r0, r1 = 0, 0.0, 1.0
    if DataKey.AMD_RBF in data:
        nb, r0, r1 = data[DataKey.AMD_RBF].tolist()
        nb = int(nb)
    out = ops.edge_geom(
        data[DataKey.POSITIONS], data[DataKey.EDGE_INDEX], data.get(DataKey.EDGE_CELL_SHIFT),
        data.get(DataKey.CELL), data.get(DataKey.BATCH), data[DataKey.AMD_PERM], lmax, nb, r0, r1,
        want_vectors=want_vectors, want_lengths=want_lengths, want_attrs=want_attrs,
        want_embedding=want_embedding and nb > 0,
    )
    data[DataKey.AMD_GEOM], data[DataKey.AMD_SH] = out["geom_sorted"], out["sh_sorted"]
    for key, name in ((DataKey.EDGE_VECTORS, "edge_vectors"), (DataKey.EDGE_LENGTH, "edge_lengths"),
                      (DataKey.EDGE_ATTRS, "edge_attrs"), (DataKey.EDGE_EMBEDDING, "edge_embedding")):
        if out[name] is not None:
            data[key] = out[name]
    return data


def ensure_training_edge_tensors(data: DataKey.Type) -> DataKey.Type:
    """Extra per-edge tensor the tensor-product adjoint needs (built once per batch): destination ids in
    destination-sorted order."""
    if "_amd_dst_sorted" not in data:
        ensure_edge_geometry(data)
        perm = data[DataKey.AMD_PERM].long()
        data["_amd_dst_sorted"] = data[DataKey.EDGE_INDEX][1][perm].to(torch.int32).contiguous()
        # the sorted edges grouped by SOURCE node (stable): the order in which dL/dx is summed per node -- the CSR builder
        # keyed on the source column of the destination-sorted list (its "source" output is not used)
        src = data[DataKey.AMD_SRC].long()
        pairs = torch.stack([src, src])
        out_perm, out_ptr, _, _ = ops.csr_build(pairs, int(data[DataKey.POSITIONS].shape[0]))
        data["_amd_out_csr"] = (out_ptr, out_perm)
    return data


def with_edge_vectors(data: DataKey.Type, with_lengths: bool = True) -> DataKey.Type:
    return ensure_edge_geometry(data, want_vectors=True, want_lengths=with_lengths)


class SphericalHarmonicEdgeAttrs(ModuleIrreps, torch.nn.Module):
    out_field: str

    def __init__(
        self,
        irreps_edge_sh: Union[int, str, Irreps],
        edge_sh_normalization: str = "component",
        edge_sh_normalize: bool = True,
        irreps_in=None,
        out_field: str = DataKey.EDGE_ATTRS,
        materialize: bool = False,
    ):
        super().__init__()
        self.out_field = out_field
        if isinstance(irreps_edge_sh, int):
            self.irreps_edge_sh = Irreps.spherical_harmonics(irreps_edge_sh)
        else:
            self.irreps_edge_sh = Irreps(irreps_edge_sh)
        lmax = len(self.irreps_edge_sh) - 1
        if self.irreps_edge_sh != Irreps.spherical_harmonics(lmax) or lmax > 4:
            raise NotImplementedError(f"edge SH must be 0e+1o+...+lmax (lmax<=4), got {self.irreps_edge_sh}")
        if edge_sh_normalization != "component" or not edge_sh_normalize:
            raise NotImplementedError("only normalize=True, normalization='component' (the reference default)")
        self.lmax = lmax
        self.materialize = materialize
        self.init_irreps(irreps_in=irreps_in, irreps_out={out_field: self.irreps_edge_sh})

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        ensure_edge_geometry(data, lmax=self.lmax, want_vectors=self.materialize, want_attrs=self.materialize)
        if self.materialize and self.out_field != DataKey.EDGE_ATTRS:
            data[self.out_field] = data[DataKey.EDGE_ATTRS]
        return data
