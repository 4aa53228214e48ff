"""
Point convolution of the tensor field network on MI355X (mirrors reference nn/conv.py:26-215).

  node_self_connection = sc(x, species)
  x1   = lin1(x, species)
  agg  = sum_{edges -> n} uvu(x1[src], Y(edge), radial_mlp(|edge|)) / sqrt(avg_num_neighbors)
  out  = lin2(agg, species) + node_self_connection        [-> Gate -> BatchNorm]
"""
from typing import Dict

import torch

import numpy as np

from .. import autograd as _ag
from .. import ops
from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irreps
from ._tables import DerivedWeight, DeviceTables
from .utils import ActivationLayer, NormalizationLayer, SpeciesLinear, UVUTensorProduct


class PointConv(ModuleIrreps, torch.nn.Module):
    def __init__(
        self,
        irreps_in: Dict[str, Irreps],
        conv_layer_irreps: Irreps,
        fc_num_hidden_layers: int = 1,
        fc_hidden_size: int = 8,
        avg_num_neighbors: int = None,
    ):
        super().__init__()
        self.avg_num_neighbors = avg_num_neighbors
        self.init_irreps(irreps_in)
        feats_in = self.irreps_in[DataKey.NODE_FEATURES]
        n_species = self.irreps_in[DataKey.NODE_ATTRS].dim
        edge_attrs = self.irreps_in[DataKey.EDGE_ATTRS]
        conv_layer_irreps = Irreps(conv_layer_irreps)

        self.lin1 = SpeciesLinear(feats_in, n_species, feats_in)
        self.tp = UVUTensorProduct(
            feats_in,
            edge_attrs,
            conv_layer_irreps,
            mlp_input_size=self.irreps_in[DataKey.EDGE_EMBEDDING].dim,
            mlp_hidden_size=fc_hidden_size,
            mlp_num_hidden_layers=fc_num_hidden_layers,
            mlp_activation="silu",
        )
        # the uvu product only emits the reachable paths, so lin2 reads tp.irreps_out
        self.lin2 = SpeciesLinear(self.tp.irreps_out, n_species, conv_layer_irreps)
        self.sc = SpeciesLinear(feats_in, n_species, conv_layer_irreps)
        self.irreps_out[DataKey.NODE_FEATURES] = conv_layer_irreps
        self._lin1_sc = None  # built at first inference forward
        self._lin1_sc_packed = DerivedWeight(self._pack_lin1_sc)

    def _pack_lin1_sc(self, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
        return torch.cat([self.lin1._pack(w1), self.sc._pack(w2)], dim=1).contiguous()

    def _fused_lin1_sc_tables(self):
        """lin1 and the self-connection read the same rows with the same species order: in inference they run as ONE
        matten_species_linear launch writing [x1 | self_connection] side by side (segment tables concatenated, the
        self-connection's weight and output offsets shifted).  The parameters stay two reference-layout tensors."""
        if self._lin1_sc is None:
            p1, p2 = self.lin1.plan, self.sc.plan
            if len(p1.passes) != 1 or len(p2.passes) != 1 or not (p1.fully_covered and p2.fully_covered):
                self._lin1_sc = False
            else:
                seg2 = p2.passes[0].copy()
                seg2[:, 3] += p1.w_stride
                seg2[:, 5] += p1.d_out
                segs = np.concatenate([p1.passes[0], seg2])
                segs = segs[np.argsort(segs[:, 0], kind="stable")]  # blocks reading the same input columns adjacent
                self._lin1_sc = DeviceTables(meta=segs)
        return self._lin1_sc

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        x = data[DataKey.NODE_FEATURES]
        species = data[DataKey.AMD_SPECIES]
        fused = None if _ag.needs_grad(x, self.lin1.weight, self.sc.weight) else self._fused_lin1_sc_tables()
        if fused:
            wp = self._lin1_sc_packed.get(self.lin1.weight, self.sc.weight)
            d1, d2 = self.lin1.plan.d_out, self.sc.plan.d_out
            both = ops.species_linear(x, species, wp, self.lin1.plan.w_stride + self.sc.plan.w_stride,
                                      [fused.get("meta", x.device)], d1 + d2)
            x1, self_connection = both[:, :d1], both[:, d1:]
        else:
            self_connection = self.sc(x, species)
            x1 = self.lin1(x, species)
        agg = self.tp(x1, data, self.avg_num_neighbors)
        data[DataKey.NODE_FEATURES] = self.lin2(agg, species, add=self_connection)
        return data


class PointConvWithActivation(ModuleIrreps, torch.nn.Module):
    def __init__(
        self,
        irreps_in: Dict[str, Irreps],
        conv_layer_irreps: Irreps,
        fc_num_hidden_layers: int = 1,
        fc_hidden_size: int = 8,
        avg_num_neighbors: int = None,
        activation_type: str = "gate",
        activation_scalars: Dict[str, str] = {"e": "silu", "o": "tanh"},
        activation_gates: Dict[str, str] = {"e": "sigmoid", "o": "tanh"},
        normalization: str = None,
    ):
        super().__init__()
        self.init_irreps(irreps_in)
        self.act = ActivationLayer(
            self.irreps_in[DataKey.NODE_FEATURES],
            self.irreps_in[DataKey.EDGE_ATTRS],
            Irreps(conv_layer_irreps),
            activation_type=activation_type,
            activation_scalars=activation_scalars,
            activation_gates=activation_gates,
        )
        self.conv = PointConv(
            irreps_in=self.irreps_in,
            conv_layer_irreps=self.act.irreps_in,
            fc_num_hidden_layers=fc_num_hidden_layers,
            fc_hidden_size=fc_hidden_size,
            avg_num_neighbors=avg_num_neighbors,
        )
        self.norm = NormalizationLayer(self.act.irreps_out, method=normalization)
        self.irreps_out[DataKey.NODE_FEATURES] = self.act.irreps_out

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data = self.conv(data)
        # Gate and (eval-mode) BatchNorm run as one elementwise kernel
        data[DataKey.NODE_FEATURES] = self.act(data[DataKey.NODE_FEATURES], self.norm)
        return data
