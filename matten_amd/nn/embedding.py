"""
Input embeddings of the backbone on MI355X.

  SpeciesEmbedding     mirrors reference nn/embedding.py:12-110: atomic number -> species index
                       (LUT) -> one-hot -> Linear.  The one-hot times Linear is a column lookup, so
                       the kernel never builds the [N,S] one-hot unless asked to.
  EdgeLengthEmbedding  mirrors reference nn/embedding.py:158-203 (Bessel basis * sqrt(num_basis)).
                       The radial MLP recomputes the basis in its prologue, so this module only
                       records the basis parameters (and materialises the tensors on request).
"""
from typing import Dict, List, Tuple

import numpy as np
import torch

from .. import ops
from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irreps


class _AtomicNumberToIndex(torch.nn.Module):
    """Non-consecutive atomic numbers -> consecutive species indices (reference nn/embedding.py:206-263)."""

    def __init__(self, allowed_atomic_numbers: List[int]):
        super().__init__()
        allowed = torch.as_tensor(sorted(allowed_atomic_numbers), dtype=torch.long)
        lut = torch.full((int(allowed.max() - allowed.min()) + 1,), -1, dtype=torch.long)
        lut[allowed - allowed.min()] = torch.arange(len(allowed), dtype=torch.long)
        self.register_buffer("_min_Z", allowed.min())
        self.register_buffer("_max_Z", allowed.max())
        self.register_buffer("_num_species", torch.as_tensor(len(allowed)))
        self.register_buffer("_Z_to_index", lut)
        # host copies: kernel arguments must not force a device sync
        self._host = (int(allowed.min()), int(allowed.max()), len(allowed))
        self._allowed = [int(z) for z in allowed]

    @property
    def num_species(self):
        return self._num_species

    def raise_for_flags(self, flags: int, atomic_numbers: torch.Tensor):
        """Same messages as the reference's forward (nn/embedding.py:238-257)."""
        if flags & 2:
            raise RuntimeError(
                "Invalid atomic numbers. Expect atomic numbers to be in the range "
                f"[{self._host[0]}, {self._host[1]}], but got min {int(atomic_numbers.min())} "
                f"and max {int(atomic_numbers.max())}"
            )
        if flags & 4:
            z = atomic_numbers.cpu()
            for i, n in enumerate(z.tolist()):
                if n not in self._allowed:
                    raise RuntimeError(
                        f"Expect atomic numbers to be in {self._allowed}, "
                        f"got invalid atomic numbers `{n}` for data point `{i}`."
                    )

    def forward(self, atomic_numbers: torch.Tensor) -> torch.Tensor:
        # standalone use (the reference's unit test): pure index arithmetic, any device
        lo, hi, _ = self._host
        if atomic_numbers.min() < lo or atomic_numbers.max() > hi:
            self.raise_for_flags(2, atomic_numbers)
        index = self._Z_to_index[atomic_numbers - lo]
        if index.min() < 0:
            self.raise_for_flags(4, atomic_numbers)
        return index


class SpeciesEmbedding(ModuleIrreps, torch.nn.Module):
    def __init__(
        self,
        irreps_in: Dict[str, Irreps] = None,
        embedding_dim: int = 16,
        num_species: int = None,
        allowed_species: List[int] = None,
        out_fields: Tuple[str] = (DataKey.NODE_ATTRS, DataKey.NODE_FEATURES),
        use_atom_feats: bool = False,
        atom_feats_dim: int = None,
        materialize: bool = False,
        check_species: bool = True,
    ):
        super().__init__()
        if allowed_species is not None and num_species is not None:
            raise ValueError("allowed_species and num_species cannot both be provided.")
        if allowed_species is None:
            raise NotImplementedError("matten_amd needs `allowed_species` (every shipped config provides it)")
        if use_atom_feats and atom_feats_dim is None:
            raise ValueError("`atom_feats_dim` must be provided if `use_atom_feats` is True.")
        self.use_atom_feats = use_atom_feats
        self.embedding_dim = embedding_dim
        self.out_fields = out_fields
        self.materialize = materialize
        self.check_species = check_species
        self.atomic_number_to_index = _AtomicNumberToIndex(allowed_species)
        self.num_species = len(allowed_species)
        self.init_irreps(
            irreps_in,
            {DataKey.NODE_ATTRS: Irreps(f"{self.num_species}x0e"),
             # reference nn/embedding.py:59-68: the per-atom input features ride behind the species embedding
             DataKey.NODE_FEATURES: Irreps(f"{embedding_dim + (atom_feats_dim if use_atom_feats else 0)}x0e")},
        )
        self.linear = torch.nn.Linear(self.num_species, embedding_dim)

    def finish_checks(self) -> None:
        """check_species == "deferred": wait for and evaluate the validation flags of the most recent forward"""
        pending, self._pending = getattr(self, "_pending", None), None
        if pending is None:
            return
        ev, host, Z, n_nodes = pending
        ev.synchronize()
        self._raise_for(host.tolist(), Z, n_nodes)

    def _raise_for(self, flags, Z, n_nodes) -> None:
        if flags[0]:
            self.atomic_number_to_index.raise_for_flags(flags[0], Z)
        if flags[1] & 1:
            rng = f"[0, {n_nodes})" if n_nodes is not None else "the batch's node range"
            raise IndexError(f"edge_index holds node ids outside {rng} (a malformed batch)")

    def raise_for_last_flags(self, n_nodes=None) -> None:
        """host sync: read the validation flags of the most recent forward and raise what the reference would have
        (unknown species: RuntimeError / ValueError of _AtomicNumberToIndex; edge_index out of range: IndexError)"""
        flags_dev = getattr(self, "_last_flags", None)
        if flags_dev is None:
            return
        self._raise_for(flags_dev.tolist(), self._last_Z, n_nodes)

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        a2i = self.atomic_number_to_index
        lo, hi, S = a2i._host
        if DataKey.SPECIES_INDEX in data:
            # already indexed: feed indices through an identity LUT
            Z = data[DataKey.SPECIES_INDEX]
            lut = torch.arange(S, dtype=torch.int64, device=Z.device)
            lo, hi = 0, S - 1
            provided = True
        elif DataKey.ATOMIC_NUMBERS in data:
            Z, lut, provided = data[DataKey.ATOMIC_NUMBERS], a2i._Z_to_index, False
        else:
            raise ValueError("Nothing in `data` to encode. Need either species_index or atomic_numbers")
        # one flag vector for the three kernels that validate their input: [species, edge_index range, grouping]
        flags_dev = torch.zeros(3, dtype=torch.int32, device=Z.device)
        sidx, s32, feats, attrs, _ = ops.species_embed(
            Z, lut, lo, hi, S, self.linear.weight, self.linear.bias, want_attrs=self.materialize, err=flags_dev[0:1]
        )
        # the destination-sorted CSR every conv layer walks is built here, so that its range check of edge_index
        # shares this module's one host sync (the reference would raise an IndexError in its first gather,
        # nn/_nequip.py:238; the kernels clamp, so a malformed batch must not get past this point)
        if DataKey.EDGE_INDEX in data:
            from ._nequip import ensure_graph

            built_here = DataKey.AMD_ROWPTR not in data
            csr_err = ensure_graph(data, err=flags_dev[1:2]).get("_amd_csr_err")
            if not built_here and csr_err is not None:   # built earlier by another module: it has its own flag word
                flags_dev[1:2] = csr_err
        # nodes grouped by species (stable): the species-indexed linears walk this order.  Enqueued before the host
        # waits for the flags below, so that its launches are not paced by the host afterwards (an unknown species
        # is grouped with species 0 by the kernel and never used: the check raises)
        order, seg, _ = ops.group_by_key(sidx, S, err=flags_dev[2:3])
        # kept for a deferred check (matten_amd.graphs: inside a captured graph this is a static tensor every replay
        # rewrites, read on request after the replay)
        self._last_flags, self._last_Z = flags_dev, Z
        n_nodes = data[DataKey.POSITIONS].shape[0] if DataKey.POSITIONS in data else None
        if self.check_species == "deferred":
            # Pipelined validation: the flags of THIS forward travel to pinned host memory behind an event and are read
            # when the NEXT forward (or finish_checks()) comes by -- the host never waits for the device it has just
            # fed, so a loop of forwards runs without the launch gaps the immediate check leaves behind its sync.
            # A malformed batch therefore raises one forward late (the kernels clamp: nothing unsafe happens meanwhile).
            self.finish_checks()
            host = torch.empty(3, dtype=torch.int32, pin_memory=True)
            host.copy_(flags_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(Z.device))
            self._pending = (ev, host, Z, n_nodes)
        elif self.check_species:
            self.raise_for_last_flags(n_nodes)
        if not provided:
            data[DataKey.SPECIES_INDEX] = sidx
        data[DataKey.AMD_SPECIES] = (order, seg)
        data[DataKey.AMD_SPECIES_I32] = s32
        if attrs is not None:
            data[DataKey.NODE_ATTRS] = attrs
        if torch.is_grad_enabled() and self.linear.weight.requires_grad:
            # training path: the same kernel output; the adjoint sums the gradient per species (autograd.SpeciesEmbedFn)
            from ..autograd import SpeciesEmbedFn

            if self.__dict__.get("_wgrad_segs") is None:
                from ._tables import DeviceTables

                dim = self.linear.weight.shape[0]
                self.__dict__["_wgrad_segs"] = DeviceTables(segs=np.array([[0, 1, 1, 0, dim, 0, 0, 0]], dtype=np.int32))
            feats = SpeciesEmbedFn.apply(self.linear.weight, self.linear.bias, feats, (order, seg),
                                         self._wgrad_segs.get("segs", feats.device))
        if self.use_atom_feats:
            # reference nn/embedding.py:103-105: torch.hstack((embed, data["atom_feats"])) -- a copy, no arithmetic
            extra = data["atom_feats"]
            if extra.dim() != 2 or extra.shape[0] != feats.shape[0]:
                raise ValueError(f"atom_feats must be [n_atoms, atom_feats_dim], got {tuple(extra.shape)}")
            feats = torch.cat([feats, extra.to(feats.dtype)], dim=1)
        data[DataKey.NODE_FEATURES] = feats
        return data


class EdgeLengthEmbedding(ModuleIrreps, torch.nn.Module):
    REQUIRED_KEYS_IRREPS_IN = [DataKey.POSITIONS, DataKey.EDGE_INDEX]

    def __init__(
        self,
        irreps_in: Dict[str, Irreps] = None,
        out_field: str = DataKey.EDGE_EMBEDDING,
        num_basis: int = 10,
        start: float = 0.0,
        end: float = 5.0,
        basis: str = "bessel",
        cutoff: bool = True,
        materialize: bool = False,
    ):
        super().__init__()
        if basis != "bessel" or not cutoff:
            raise NotImplementedError("matten_amd implements the Bessel basis with cutoff (all shipped configs)")
        self.num_basis, self.start, self.end, self.basis, self.cutoff = num_basis, start, end, basis, cutoff
        self.out_field = out_field
        self.materialize = materialize
        self.init_irreps(irreps_in, irreps_out={out_field: Irreps(f"{num_basis}x0e")})

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        from ._nequip import ensure_edge_geometry

        data[DataKey.AMD_RBF] = torch.tensor([self.num_basis, self.start, self.end], dtype=torch.float64)
        if self.materialize:
            ensure_edge_geometry(data, want_lengths=True, want_embedding=True)
        return data
