"""
Point convolution of the tensor field network on MI355X (mirrors reference nn/conv.py:26-215).

  node_self_connection = sc(x, species)
  x1   = lin1(x, species)
  agg  = sum_{edges -> n} uvu(x1[src], Y(edge), radial_mlp(|edge|)) / sqrt(avg_num_neighbors)
  out  = lin2(agg, species) + node_self_connection        [-> Gate -> BatchNorm]
"""
from typing import Dict

import torch

from ..data.irreps import DataKey, ModuleIrreps
from ..o3 import Irreps
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

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        x = data[DataKey.NODE_FEATURES]
        species = data[DataKey.AMD_SPECIES]
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
