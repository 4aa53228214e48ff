"""
Equivariant building blocks of the conv layer on MI355X (mirrors reference nn/utils.py).

  SpeciesLinear        stands in for e3nn ``FullyConnectedTensorProduct(x, one_hot(species))``
                       (reference nn/conv.py:59-61,77-79,84-86) and, with one species, ``o3.Linear``
  RadialMLP            e3nn ``FullyConnectedNet`` of the radial weights (reference nn/utils.py:246-251)
  UVUTensorProduct     reference nn/utils.py:170-277 -- here fused with the gather of x[src], the
                       scatter onto the destination node and the neighbour normalisation
  ActivationLayer      reference nn/utils.py:29-167 (Gate)
  NormalizationLayer   reference nn/utils.py:397-437 (BatchNorm)

Parameters keep the reference's names, shapes and flat layouts.
"""
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import autograd as _ag
from .. import ops, plan as _plan
from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irrep, Irreps
from ._activation import act_const_table, normalize2mom_const
from ._tables import DerivedWeight, DeviceTables

# activation *names* per parity class (reference nn/utils.py:14-26 holds the callables)
ACTIVATION = {"e": {"ssp": "ssp", "silu": "silu", "sigmoid": "sigmoid"}, "o": {"abs": "abs", "tanh": "tanh"}}

tp_path_exists = _plan.tp_path_exists


def _inverse_permutation(gather):
    """flat parameter index -> position in the packed [S, w_stride] table (gather is a bijection, see plan.py)"""
    import numpy as np

    g = np.asarray(gather).reshape(-1)
    assert g.size == 0 or (np.sort(g) == np.arange(g.size)).all(), "packed weight table is not a permutation"
    inv = np.empty_like(g)
    inv[g] = np.arange(g.size, dtype=g.dtype)
    return inv


class SpeciesLinear(torch.nn.Module):
    """out[n] = sum_u W[u, species(n), w] x[n,u] / sqrt(fan_in), per irrep; flat ``weight`` as in e3nn."""

    def __init__(self, irreps_in, n_species: Optional[int], irreps_out):
        super().__init__()
        if n_species is None:
            self.plan = _plan.plan_linear(irreps_in, irreps_out)
        else:
            self.plan = _plan.plan_fctp(irreps_in, n_species, irreps_out)
        self.irreps_in, self.irreps_out = self.plan.irreps_in, self.plan.irreps_out
        self.n_species = n_species
        self.weight = torch.nn.Parameter(torch.randn(self.plan.weight_numel))
        self._tables = DeviceTables(
            gather=self.plan.gather, scale=self.plan.scale, perm_t=self.plan.perm_t,
            gather_inv=_inverse_permutation(self.plan.gather),
            **{f"meta{i}": m for i, m in enumerate(self.plan.passes)},
            **{f"meta_t{i}": m for i, m in enumerate(self.plan.passes_t)},
        )
        self._packed = DerivedWeight(self._pack)

    def _pack(self, weight: Tensor) -> Tensor:
        dev = weight.device
        return (weight[self._tables.get("gather", dev)] * self._tables.get("scale", dev)).contiguous()

    def forward(self, x: Tensor, species_order=None, add: Optional[Tensor] = None) -> Tensor:
        """species_order = (order, seg): node ids sorted by species + per-species offsets (DataKey.AMD_SPECIES)."""
        if self.n_species is not None and species_order is None:
            raise ValueError("species order required")
        order = species_order if self.n_species is not None else None
        if _ag.needs_grad(x, self.weight, add):
            # training path: packing, linear and both adjoints are ONE autograd node (autograd.SpeciesLinearFn)
            return _ag.SpeciesLinearFn.apply(x, self.weight, add, self, order)
        wp = self._packed.get(self.weight)
        metas = [self._tables.get(f"meta{i}", x.device) for i in range(len(self.plan.passes))]
        return ops.species_linear(x, species_order if self.n_species is not None else None, wp, self.plan.w_stride,
                                  metas, self.plan.d_out, add, self.plan.fully_covered)


class _RadialLayer(torch.nn.Module):
    def __init__(self, h_in: int, h_out: int):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(h_in, h_out))


# radial MLPs that read the same edge lengths (one per conv layer of a model), by group token.  The modules only carry
# the integer token (picklable, copyable); a copied / unpickled module is not in the registry and simply runs alone.
_RADIAL_GROUPS: Dict[int, list] = {}


def link_radial_group(mlps) -> None:
    import weakref

    token = max(_RADIAL_GROUPS, default=0) + 1
    _RADIAL_GROUPS[token] = [weakref.ref(m) for m in mlps]
    for m in mlps:
        m._radial_group = token


def radial_group_of(mlp) -> list:
    members = [m for m in (r() for r in _RADIAL_GROUPS.get(getattr(mlp, "_radial_group", 0), ())) if m is not None]
    return members if any(m is mlp for m in members) else [mlp]


class RadialMLP(torch.nn.Module):
    """Bias-free MLP [n_basis, h, h, W]: x <- c*silu(x @ W/sqrt(h_in)) on hidden layers."""

    def __init__(self, hs: List[int], act: str = "silu", out_cols=None):
        """out_cols (int64 array, optional): emit output column j = reference column out_cols[j] (-1: zeros).
        The state_dict keeps the reference layout; only the packed copy the kernel reads is permuted."""
        super().__init__()
        self.out_cols = None if out_cols is None else torch.as_tensor(out_cols, dtype=torch.int64)
        if len(hs) != 4 or hs[1] != 32 or hs[2] != 32:
            raise NotImplementedError(f"radial MLP must be [nb, 32, 32, W] (invariant_layers=2, neurons=32); got {hs}")
        if act != "silu":
            raise NotImplementedError("radial MLP activation must be silu (reference nn/conv.py:72)")
        self.hs = list(hs)
        self.act_cst = normalize2mom_const(act)
        for i, (a, b) in enumerate(zip(hs, hs[1:])):
            setattr(self, f"layer{i}", _RadialLayer(a, b))
        self._packed = DerivedWeight(self._pack)
        self._h_scale = DerivedWeight(self._fp16_scale)
        self._h_scale_c = None

    def _fp16_scale(self, w0: Tensor, w1: Tensor) -> Tensor:
        """[s, 1/s]: the power of two s <= 1 the hidden features are multiplied by before their fp16 hi/lo split
        (csrc/tp_fused.hip) so that |s h2| < 2^15 for EVERY possible edge length -- a bound, not a measurement, so no
        host sync and no data dependence:  |bessel_k| <= sqrt(2/c) sqrt(nb) pi (k+1) / c  (sin(x)/x <= 1),
        |silu(z)| <= |z|, hence |h2| <= P(c) * max_col sum_k |W0p[k,col]| (k+1) * max_col sum_k |W1p[k,col]|.
        s = 1 whenever the bound is below 2^15 (every normally scaled MLP: the results are then bit-identical to an
        unscaled run); a checkpoint whose radial weights are orders of magnitude larger gets s < 1 instead of inf."""
        nb, h = self.hs[0], self.hs[1]
        c = self._h_scale_c
        k1 = torch.arange(1, nb + 1, device=w0.device, dtype=w0.dtype)[:, None]
        n0 = ((w0.abs() / nb**0.5) * k1).sum(0).max()
        n1 = (w1.abs() * (self.act_cst / h**0.5)).sum(0).max()
        bound = (2.0 / c) ** 0.5 * nb**0.5 * 3.141592653589793 / c * n0 * n1
        s = torch.where(bound > 2.0**15, torch.exp2(torch.floor(torch.log2(2.0**14 / bound.clamp(min=1e-30)))),
                        torch.ones_like(bound))
        s = torch.where(torch.isfinite(s) & (s > 0), s, torch.full_like(s, 2.0**-100))
        return torch.stack([s, 1.0 / s]).float().contiguous()

    def h_scale(self, r_start: float, r_end: float) -> Tensor:
        c = float(r_end - r_start)
        if c != self._h_scale_c:
            self._h_scale_c = c
            self._h_scale._key = None
        return self._h_scale.get(self.layer0.weight, self.layer1.weight)

    def _pack(self, w0: Tensor, w1: Tensor, w2: Tensor):
        nb, h, W = self.hs[0], self.hs[1], self.hs[3]
        w2s = w2 * (self.act_cst / h**0.5)
        if self.out_cols is not None:
            cols = self.out_cols.to(w2.device)
            w2s = torch.where(cols[None, :] >= 0, w2s[:, cols.clamp(min=0)], w2s.new_zeros(()))
            W = cols.numel()
        # +16 spare columns: the fused kernel reads whole 16-column tiles starting at any entry
        nb_pad, w_pad = (nb + 3) // 4 * 4, (W + 15) // 16 * 16 + 16
        w0p = w0.new_zeros(nb_pad, h)
        w0p[:nb] = w0 / nb**0.5
        w1p = (w1 * (self.act_cst / h**0.5)).contiguous()
        w2p = w2.new_zeros(h, w_pad)
        w2p[:, :W] = w2s
        return w0p, w1p, w2p

    def pack_last(self, w2: Tensor, cols: Tensor) -> Tensor:
        """the last layer pre-scaled with its columns in ANOTHER consumer's order (cols[j] = reference column of output
        column j, -1 = zeros; e.g. the conv-tile kernel's entries), padded like the packed copy of _pack"""
        h = self.hs[1]
        w2s = w2 * (self.act_cst / h**0.5)
        cols = cols.to(w2.device)
        w2s = torch.where(cols[None, :] >= 0, w2s[:, cols.clamp(min=0)], w2s.new_zeros(()))
        W = cols.numel()
        w2p = w2.new_zeros(h, (W + 15) // 16 * 16 + 16)
        w2p[:, :W] = w2s
        return w2p

    def forward(self, geom_sorted: Tensor, n_basis: int, r_start: float, r_end: float) -> Tensor:
        """Per-edge weights w[E, w_pad] (all three layers)."""
        if n_basis != self.hs[0]:
            raise ValueError(f"radial basis size {n_basis} != MLP input {self.hs[0]}")
        w0p, w1p, w2p = self._packed.get(self.layer0.weight, self.layer1.weight, self.layer2.weight)
        return ops.radial_mlp(geom_sorted, n_basis, r_start, r_end, w0p, w1p, w2p)

    def pack_scales(self):
        """(1/sqrt(nb), c/sqrt(h), c/sqrt(h)): what turns the raw layers into the kernels' operands"""
        nb, h = self.hs[0], self.hs[1]
        return 1.0 / nb**0.5, self.act_cst / h**0.5, self.act_cst / h**0.5

    def pack_reference_order(self, w0: Tensor, w1: Tensor, w2: Tensor):
        """(w0p, w1p, w2p) with the LAST layer's columns in the reference's order (what the training tensor product and
        its adjoint index), padded to a multiple of 16"""
        nb, h, W = self.hs[0], self.hs[1], self.hs[3]
        nb_pad, w_pad = (nb + 3) // 4 * 4, (W + 15) // 16 * 16
        w0p = w0.new_zeros(nb_pad, h)
        w0p[:nb] = w0 / nb**0.5
        w1p = (w1 * (self.act_cst / h**0.5)).contiguous()
        w2p = w2.new_zeros(h, w_pad)
        w2p[:, :W] = w2 * (self.act_cst / h**0.5)
        return w0p, w1p, w2p

    def forward_train(self, geom_sorted: Tensor, n_basis: int, r_start: float, r_end: float) -> Tensor:
        """w[E, w_pad] in the reference column order, differentiable w.r.t. the three weight matrices: forward and
        adjoint are this library's MFMA kernels (matten_radial_mlp / matten_radial_mlp_bwd).  Same arithmetic as e3nn
        FullyConnectedNet: x <- c*silu(x @ W/sqrt(h_in)); last layer linear."""
        if n_basis != self.hs[0]:
            raise ValueError(f"radial basis size {n_basis} != MLP input {self.hs[0]}")
        return _ag.RadialMLPFn.apply(self.layer0.weight, self.layer1.weight, self.layer2.weight, self, geom_sorted,
                                     n_basis, r_start, r_end)

    def hidden(self, geom_sorted: Tensor, n_basis: int, r_start: float, r_end: float, data=None):
        """(h2s[E,2,32] scaled by h_scale(), w2p): the two hidden layers evaluated, the last layer left to the fused TP kernel.
        With `data` (the batch dict) and a sibling group (set by the model factory: every conv layer's radial MLP reads
        the same edge lengths) the first call evaluates ALL siblings in one launch and parks the results in the dict."""
        if n_basis != self.hs[0]:
            raise ValueError(f"radial basis size {n_basis} != MLP input {self.hs[0]}")
        w0p, w1p, w2p = self._packed.get(self.layer0.weight, self.layer1.weight, self.layer2.weight)
        src = self.__dict__.get("_hidden_from")
        if src is not None:   # an inference view (nn/conv.py): the hidden layers ARE the full layer's (shared weights)
            return src.hidden(geom_sorted, n_basis, r_start, r_end, data)[0], w2p
        group = radial_group_of(self)
        if data is not None and len(group) > 1 and os.environ.get("MATTEN_RADIAL_MULTI", "1") != "0":
            cache = data.get("_amd_h2s")
            if cache is None or cache.get("geom") is not geom_sorted or id(self) not in cache:
                packs = [m._packed.get(m.layer0.weight, m.layer1.weight, m.layer2.weight) for m in group]
                outs = ops.radial_hidden_multi(geom_sorted, n_basis, r_start, r_end, [p[0] for p in packs],
                                               [p[1] for p in packs], [m.h_scale(r_start, r_end) for m in group])
                cache = {"geom": geom_sorted, **{id(m): h for m, h in zip(group, outs)}}
                data["_amd_h2s"] = cache
            return cache.pop(id(self)), w2p   # popped: the 147 MB per layer are released after use
        return ops.radial_hidden(geom_sorted, n_basis, r_start, r_end, w0p, w1p, self.h_scale(r_start, r_end)), w2p


# small batches: CSR segments longer than the piece length are walked in pieces (ops.csr_split); 0 = off.  Large batches
# (>= HUB_SPLIT_MAX_ROWS nodes) keep whole segments: regular degrees, and the pieces' rows would cost memory traffic.
# The best length grows with the batch (hipGraph forward in ms, tools/dbg/hub_len_sweep*.py; pieces are balanced: 18 edges
# at length 16 are 9 + 9):  n100 tiled 1 / 3 / 10 / 17 times   8: 0.455 0.637 1.350 2.092   16: 0.480 0.611 1.114 1.677
# 32: 0.575 0.660 1.079 1.549   whole segments: 0.877 0.985 1.378 1.844;   fcc-64, 4 / 11 / 30 / 60 / 120 crystals
# 8: 0.365 0.433 0.584 0.850 1.473   16: 0.368 0.418 0.544 0.754 1.242   32 (= whole): 0.408 0.453 0.540 0.693 1.077
# -- few nodes need the parallelism, many pay for the pieces' rows.
_HUB_SPLIT_LEN_ENV = os.environ.get("MATTEN_HUB_SPLIT_LEN")
HUB_SPLIT_LEN = int(_HUB_SPLIT_LEN_ENV) if _HUB_SPLIT_LEN_ENV is not None else 16   # the training forward's length (autograd.py)


def hub_split_len(n_rows: int) -> int:
    """piece length of the inference forward for a batch of n_rows nodes (MATTEN_HUB_SPLIT_LEN overrides)"""
    if _HUB_SPLIT_LEN_ENV is not None:
        return HUB_SPLIT_LEN
    return 8 if n_rows <= 768 else (16 if n_rows <= 3072 else 32)


HUB_SPLIT_MAX_ROWS = int(os.environ.get("MATTEN_HUB_SPLIT_MAX_ROWS", "8192"))
# training: from this many edges on the forward runs on the fused kernel (MATTEN_TRAIN_TP = fused | paths overrides)
TRAIN_FUSED_MIN_EDGES = int(os.environ.get("MATTEN_TRAIN_TP_FUSED_MIN_EDGES", "65536"))


class UVUTensorProduct(torch.nn.Module):
    def __init__(
        self,
        irreps_in1: Irreps,
        irreps_in2: Irreps,
        irreps_out: Irreps,
        *,
        internal_and_share_weights: bool = False,
        mlp_input_size: int = None,
        mlp_hidden_size: int = 8,
        mlp_num_hidden_layers: int = 1,
        mlp_activation: str = "ssp",
    ):
        super().__init__()
        if internal_and_share_weights:
            raise NotImplementedError("internal shared weights are not used by the model factories")
        assert mlp_input_size is not None, (
            "Expect `mlp_input_size` be provided when `internal_and_share_weights` is set to `False`, got `None`"
        )
        self.plan = _plan.plan_uvu(irreps_in1, irreps_in2, irreps_out)
        self.irreps_mid = self.plan.irreps_mid
        self.weight_numel = self.plan.weight_numel
        layer_sizes = [mlp_input_size] + mlp_num_hidden_layers * [mlp_hidden_size] + [self.weight_numel]
        # "fused" : production -- last radial-MLP layer on the matrix cores inside the TP kernel, the per-edge
        #           weights never reach memory; weight columns in [entry][u][coupling] order (plan.fused_cols)
        # "paths" : one wave per path over a materialised w[E, W] (same literals, no fusion, the reference's weight
        #           layout): the training forward, and an independent implementation for the tests
        self.impl = os.environ.get("MATTEN_TP_IMPL", "fused")
        self._train_tp_auto = None   # MATTEN_TRAIN_TP=auto: the route this module's first training batch chose
        if self.impl not in ("fused", "paths"):
            raise ValueError(f"MATTEN_TP_IMPL={self.impl!r}: 'fused' or 'paths' (the table-driven and block kernels of "
                             "rounds 1-2 were removed in round 4)")
        self.weight_nn = RadialMLP(layer_sizes, act=mlp_activation,
                                   out_cols=self.plan.fused_cols if self.impl == "fused" else None)
        self._tables = DeviceTables(
            entries=self.plan.path_entries, unit_start=self.plan.unit_start,
            gentries=self.plan.group_entries, gumap=self.plan.fused_unit_map,
            bw_col_meta=self.plan.bw_col_meta, bw_nnz_ijk=self.plan.bw_nnz_ijk, bw_nnz_c=self.plan.bw_nnz_c,
            bw_in_ptr=self.plan.bw_in_ptr, bw_in_cols=self.plan.bw_in_cols,
            bw_blocks=self.plan.bw_blocks, bw_paths=self.plan.bw_paths, bw_w_entries=self.plan.bw_w_entries,
            fused_cols=self.plan.fused_cols,
        )

        self._a_split = DerivedWeight(self._split_last_layer)
        self._a_split_c = None

    def _split_last_layer(self, w0: Tensor, w1: Tensor, w2: Tensor):
        """the last radial layer as the fp16 hi/lo MFMA fragments of matten_tp_fused, with 1 / (entry scale x hidden
        feature scale) as the per-entry output factor (rebuilt when the weights change)"""
        w2p = self.weight_nn._packed.get(w0, w1, w2)[2]
        frag, scale_inv = ops.split_a_tiles(w2p, self.plan.group_entries)
        return frag, (scale_inv * self.weight_nn._h_scale.get(w0, w1)[1]).contiguous()

    def a_split(self, r_start: float, r_end: float):
        """(fragments, a_scale_inv) for matten_tp_fused / matten_tp_lin2, consistent with weight_nn.hidden()"""
        mlp = self.weight_nn
        mlp.h_scale(r_start, r_end)   # fixes the cutoff range the bound is taken for
        c = float(r_end - r_start)
        if c != self._a_split_c:     # the folded 1 / h_scale belongs to ONE cutoff range: a new range, new fragments' factor
            self._a_split_c = c
            self._a_split._key = None
        return self._a_split.get(mlp.layer0.weight, mlp.layer1.weight, mlp.layer2.weight)

    @property
    def irreps_out(self) -> Irreps:
        return self.irreps_mid.simplify()

    def forward(self, node_feats: Tensor, data: DataKey.Type, avg_num_neighbors=None, out_layout=None) -> Tensor:
        """sum over incoming edges of TP(x[src], Y(edge), MLP(rbf(edge))), normalised; [N, d_mid].
        out_layout = (group entries with component-major output offsets, row stride): plan.AggLinearPlan (fused path only)"""
        nb, r0, r1 = data[DataKey.AMD_RBF].tolist()
        dev = node_feats.device
        avg = avg_num_neighbors if avg_num_neighbors is not None else 0.0
        num_neigh = None if avg_num_neighbors is not None else data[DataKey.NUM_NEIGH]
        if _ag.needs_grad_lazy(lambda: (node_feats, *_ag.params_of(self.weight_nn))):
            # training path: radial weights materialised in the reference layout by the MLP kernel, the tensor
            # product + neighbour sum and both adjoints are HIP kernels too
            from ._nequip import ensure_training_edge_tensors

            ensure_training_edge_tensors(data)
            mode = os.environ.get("MATTEN_TRAIN_TP", "auto")
            n_edges = data[DataKey.AMD_SRC].shape[0]
            if mode == "auto":
                # decided ONCE per module, by the first training batch it sees: the two routes round differently (fused: split-fp16
                # matrix products for w, fp32-class; paths: fp32 MFMA), and batches around the threshold must not flip a run
                # between them from step to step (INTEGRATION.md "Training numerics")
                if self._train_tp_auto is None:
                    self._train_tp_auto = "fused" if n_edges >= TRAIN_FUSED_MIN_EDGES else "paths"
                mode = self._train_tp_auto
            if self.impl == "fused" and mode == "fused":
                # forward on the production kernel, w[E, W] re-evaluated per layer inside the backward only: the same step time
                # as the materialised-w forward at batch 2048 (5.14 vs 5.15 ms) with 4 x E x W x 4 bytes less live memory
                # between the passes; small batches keep the path kernels (fewer launches: 10 % faster at batch 32)
                mlp = self.weight_nn
                return _ag.FusedTensorProductFn.apply(node_feats, mlp.layer0.weight, mlp.layer1.weight, mlp.layer2.weight,
                                                      self, data, avg, num_neigh)
            w_edge = self.weight_nn.forward_train(data[DataKey.AMD_GEOM], int(nb), r0, r1)
            return _ag.TensorProductScatterFn.apply(node_feats, w_edge, self, data, avg, num_neigh)
        if self.impl == "fused":
            h2p, w2p = self.weight_nn.hidden(data[DataKey.AMD_GEOM], int(nb), r0, r1, data)
            rowptr = data[DataKey.AMD_ROWPTR]
            split = None
            piece = hub_split_len(node_feats.shape[0])
            if out_layout is None and piece > 0 and node_feats.shape[0] < HUB_SPLIT_MAX_ROWS:
                # small batch: the launch lasts as long as its longest CSR segment (one hub node walked serially by one
                # wave): walk pieces of at most hub_split_len() edges as virtual nodes and sum them afterwards, in order
                key = ("_amd_csr_split", piece, num_neigh is None)   # (the training forward cuts at its own length)
                split = data.get(key)
                if split is None or split[3] is not rowptr:
                    split = ops.csr_split(rowptr, data[DataKey.AMD_SRC].shape[0], piece, num_neigh) + (rowptr,)
                    data[key] = split
            agg = ops.tp_fused(
                node_feats, h2p, w2p, data[DataKey.AMD_SH], rowptr if split is None else split[0], data[DataKey.AMD_SRC],
                out_layout[0] if out_layout is not None else self._tables.get("gentries", dev),
                self._tables.get("gumap", dev), len(self.plan.fused_unit_map),
                self.plan.fused_lds_floats_per_wave, out_layout[1] if out_layout is not None else self.plan.d_mid, avg,
                num_neigh if split is None else split[2], a_split=self.a_split(r0, r1),
            )
            return agg if split is None else ops.segment_reduce(agg, split[1], mean=False)
        # "paths": one wave per path over a materialised w[E, W] in the reference's column order (the training forward; kept
        # selectable for inference as an independent implementation of the same contraction)
        w_edge = self.weight_nn(data[DataKey.AMD_GEOM], int(nb), r0, r1)
        return ops.tp_paths(
            node_feats, w_edge, data[DataKey.AMD_SH], data[DataKey.AMD_ROWPTR], data[DataKey.AMD_SRC],
            self._tables.get("entries", dev), self._tables.get("unit_start", dev), self.plan.units_per_tile,
            self.plan.d_mid, avg, num_neigh,
        )


class ActivationLayer(torch.nn.Module):
    def __init__(
        self,
        tp_irreps_in1: Irreps,
        tp_irreps_in2: Irreps,
        tp_irreps_out: Irreps,
        *,
        activation_type: str = "gate",
        activation_scalars: Dict[str, str] = None,
        activation_gates: Dict[str, str] = None,
    ):
        super().__init__()
        key = {"e": 1, "o": -1}
        scal = {1: "ssp", -1: "tanh"} if activation_scalars is None else {
            key[k]: ACTIVATION[k][v] for k, v in activation_scalars.items()
        }
        gat = {1: "ssp", -1: "abs"} if activation_gates is None else {
            key[k]: ACTIVATION[k][v] for k, v in activation_gates.items()
        }
        self.activation_type = activation_type
        if activation_type == "norm":   # e3nn NormActivation (reference nn/utils.py:142-150)
            self.plan = _plan.plan_norm_act(tp_irreps_in1, tp_irreps_in2, tp_irreps_out, scal)
            self._tables = DeviceTables(chan=self.plan.chan)
            return
        if activation_type != "gate":
            supported = ("gate", "norm")
            raise ValueError(f"Support `activation_type` includes {supported}, got {activation_type}")
        self.plan = _plan.plan_gate(tp_irreps_in1, tp_irreps_in2, tp_irreps_out, scal, gat)
        self._tables = DeviceTables(meta=self.plan.meta, act_cst=act_const_table().numpy())

    @property
    def irreps_in(self) -> Irreps:
        return self.plan.irreps_in

    @property
    def irreps_out(self) -> Irreps:
        return self.plan.irreps_out

    def _forward_norm_act(self, x: Tensor, norm, data) -> Tensor:
        dev = x.device
        bn = norm.n if (norm is not None and norm.method == "batch") else None
        if _ag.needs_grad(x) or (bn is not None and bn.training) or (norm is not None and norm.method == "instance"):
            y = _ag.NormActFn.apply(x, self) if _ag.needs_grad(x) else ops.norm_act(
                x, self._tables.get("chan", dev), self.plan.act_code, self.plan.epsilon)
            if norm is not None and norm.method == "instance":
                return norm.n(y, data)
            if bn is None:
                return y
            if not bn.training:
                raise NotImplementedError("gradients through eval-mode BatchNorm: call model.train()")
            return bn.forward_train(y)
        return ops.norm_act(x, self._tables.get("chan", dev), self.plan.act_code, self.plan.epsilon,
                            *((bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps) if bn is not None else ()))

    def forward(self, x: Tensor, norm: "NormalizationLayer" = None, data=None) -> Tensor:
        dev = x.device
        if self.activation_type == "norm":
            return self._forward_norm_act(x, norm, data)
        if norm is not None and norm.method == "instance":
            # Gate, then the per-crystal normalisation (needs the batch dict for `batch` / `ptr`)
            y = _ag.GateFn.apply(x, self) if _ag.needs_grad(x) else ops.gate_bn(
                x, self._tables.get("meta", dev), self._tables.get("act_cst", dev))
            return norm.n(y, data)
        bn = norm.n if (norm is not None and norm.n is not None) else None
        if bn is not None and bn.training:
            # training: Gate, then BatchNorm with batch statistics (and the running-average update)
            y = _ag.GateFn.apply(x, self) if x.requires_grad else ops.gate_bn(
                x, self._tables.get("meta", dev), self._tables.get("act_cst", dev))
            return bn.forward_train(y)
        if _ag.needs_grad(x):
            if bn is not None:
                raise NotImplementedError("gradients through eval-mode BatchNorm: call model.train()")
            return _ag.GateFn.apply(x, self)
        return ops.gate_bn(
            x, self._tables.get("meta", dev), self._tables.get("act_cst", dev),
            *((bn.running_mean, bn.running_var, bn.weight, bn.bias) if bn is not None else (None, None, None, None)),
            eps=bn.eps if bn is not None else 1e-5,
        )


class _IrrepBatchNorm(torch.nn.Module):
    """State of e3nn ``BatchNorm(irreps)``; applied inside the fused gate kernel."""

    def __init__(self, irreps: Irreps, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.irreps = Irreps(irreps)
        self.eps, self.momentum = eps, momentum
        n_scalar = sum(m for m, ir in self.irreps if ir.is_scalar())
        n_feat = self.irreps.num_irreps
        self.register_buffer("running_mean", torch.zeros(n_scalar))
        self.register_buffer("running_var", torch.ones(n_feat))
        self.weight = torch.nn.Parameter(torch.ones(n_feat))
        self.bias = torch.nn.Parameter(torch.zeros(n_scalar))
        chan, col2chan = _plan.plan_batchnorm(self.irreps)
        self._tables = DeviceTables(chan=chan, col2chan=col2chan,
                                    scalar_chan=np.nonzero(chan[:, 2])[0].astype(np.int64))

    def forward_train(self, x: Tensor) -> Tensor:
        # e3nn: running = (1 - momentum) * running + momentum * batch -- done by the statistics kernel itself
        return _ag.BatchNormTrainFn.apply(x, self.weight, self.bias, self)


class _IrrepInstanceNorm(torch.nn.Module):
    """The reference's own InstanceNorm (nn/utils.py:448-588; "more like a graph normalization": a crystal is the
    instance, its atoms the samples): scalars (every l = 0 irrep) are centred by their mean over the crystal, every
    channel is divided by sqrt(mean over the crystal of the mean squared component + eps), times ``weight``, plus
    ``bias`` on the scalars.  No running statistics: training and evaluation compute the same thing
    (matten_instance_norm_fwd / _bwd = the BatchNorm kernels with per-crystal statistics)."""

    def __init__(self, irreps: Irreps, eps: float = 1e-5):
        super().__init__()
        self.irreps = Irreps(irreps)
        self.eps = eps
        n_scalar = sum(m for m, ir in self.irreps if ir.l == 0)
        self.weight = torch.nn.Parameter(torch.ones(self.irreps.num_irreps))
        self.bias = torch.nn.Parameter(torch.zeros(n_scalar))
        chan, col2chan = _plan.plan_batchnorm(self.irreps, odd_scalars_too=True)
        self._tables = DeviceTables(chan=chan, col2chan=col2chan,
                                    scalar_chan=np.nonzero(chan[:, 2])[0].astype(np.int64))

    def forward(self, x: Tensor, data) -> Tensor:
        from ._nequip import with_batch

        with_batch(data)
        batch = data[DataKey.BATCH]
        ptr = data.get(DataKey.PTR)
        if ptr is None:   # (one host sync: the crystal count; collated batches carry `ptr`)
            n_seg = int(batch[-1]) + 1 if batch.numel() else 1
            ptr = torch.searchsorted(batch, torch.arange(n_seg + 1, device=batch.device, dtype=batch.dtype))
        if _ag.needs_grad(x, self.weight, self.bias):
            return _ag.InstanceNormFn.apply(x, self.weight, self.bias, self, ptr, batch)
        dev = x.device
        return ops.instance_norm_fwd(x, ptr, batch, self._tables.get("col2chan", dev), self._tables.get("chan", dev),
                                     self.weight, self.bias, self.eps)[0]


class NormalizationLayer(torch.nn.Module):
    def __init__(self, irreps: Irreps, method: str = None):
        super().__init__()
        self.method = method
        supported = ("batch", "instance", "none", None)
        assert method in supported, f"Unsupported normalization {method}"
        self.n = _IrrepBatchNorm(irreps) if method == "batch" else _IrrepInstanceNorm(irreps) if method == "instance" else None


class DetectAnomaly(ModuleIrreps, torch.nn.Module):
    """Checks every tensor of the data dict its predecessor produced for NaN / Inf (reference nn/utils.py:370-394);
    inserted behind every layer by ``create_sequential_module`` when the log level is DEBUG."""

    def __init__(self, irreps_in: Dict[str, Irreps], name: str):
        super().__init__()
        self.init_irreps(irreps_in=irreps_in)
        self.name = name

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        from ..utils import detect_nan_and_inf

        for k, v in data.items():
            if v is None:
                continue
            try:
                detect_nan_and_inf(v)
            except ValueError:
                raise ValueError(f"Anomaly detected for {k} of {self.name}")
        return data
