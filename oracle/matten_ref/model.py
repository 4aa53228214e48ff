"""
CPU restatement of the reference's model assembly and readout.  ORACLE / TEST INFRASTRUCTURE.

  create_model            <- model_factory/tfn_scalar_tensor.py:103-195
  create_sequential_module<- model_factory/utils.py:13-91
  ScalarTensorOracle      <- ScalarTensorModel.init_backbone/decode, tfn_scalar_tensor.py:32-79
                              (Lightning/task plumbing of model/model.py is out of scope)
  ToCartesian             <- utils.py:110-133
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Optional

import torch

from ..e3nn_lite import o3
from ..e3nn_lite.io import CartesianTensor
from . import nn as rnn

OUT_FIELD_NAME = "my_model_output"  # tfn_scalar_tensor.py:29


def create_sequential_module(modules: "OrderedDict[str, tuple]", irreps_in=None) -> rnn.Sequential:
    names, instances = [], []
    for name, (cls_type, kwargs) in modules.items():
        ir = irreps_in if not instances else instances[-1].irreps_out
        if "irreps_in" in kwargs:
            raise ValueError(f"irreps_in for module {name} is determined automatically")
        kwargs = dict(kwargs)
        kwargs["irreps_in"] = ir
        try:
            m = cls_type(**kwargs)
        except Exception as e:
            raise RuntimeError(f"Failed instantiate module `{cls_type.__name__}` with kwargs: `{kwargs}`") from e
        names.append(name)
        instances.append(m)
    return rnn.Sequential(OrderedDict(zip(names, instances)))


def create_model(hparams: Dict[str, Any], dataset_hparams: Dict[str, Any], atomic: bool = False) -> rnn.Sequential:
    """atomic=False: model_factory/tfn_scalar_tensor.py:103-195; atomic=True: tfn_atomic_tensor.py:103-199 (the node
    head maps straight to the Cartesian-tensor irreps, no pooling)."""
    layers = {
        "one_hot": (
            rnn.SpeciesEmbedding,
            {"embedding_dim": hparams["species_embedding_dim"], "allowed_species": dataset_hparams["allowed_species"],
             "use_atom_feats": hparams.get("use_atom_feats", False),
             "atom_feats_dim": dataset_hparams.get("atom_feats_size", None)},
        ),
        "spharm_edges": (rnn.SphericalHarmonicEdgeAttrs, {"irreps_edge_sh": hparams["irreps_edge_sh"]}),
        "radial_basis": (
            rnn.EdgeLengthEmbedding,
            {
                "num_basis": hparams["num_radial_basis"],
                "start": hparams["radial_basis_start"],
                "end": hparams["radial_basis_end"],
                "basis": hparams["radial_basis_type"],
            },
        ),
    }
    num_neigh = hparams["average_num_neighbors"]
    if isinstance(num_neigh, str) and num_neigh.lower() == "auto":
        num_neigh = dataset_hparams["average_num_neighbors"]
    for i in range(hparams["num_layers"]):
        layers[f"layer{i}_convnet"] = (
            rnn.PointConvWithActivation,
            {
                "conv_layer_irreps": hparams["conv_layer_irreps"],
                "activation_type": hparams["nonlinearity_type"],
                "fc_num_hidden_layers": hparams["invariant_layers"],
                "fc_hidden_size": hparams["invariant_neurons"],
                "avg_num_neighbors": num_neigh,
                "normalization": hparams["normalization"],
            },
        )
    layers["conv_layer_last"] = (
        rnn.PointConv,
        {
            "conv_layer_irreps": hparams["conv_layer_irreps"],
            "fc_num_hidden_layers": hparams["invariant_layers"],
            "fc_hidden_size": hparams["invariant_neurons"],
            "avg_num_neighbors": num_neigh,
        },
    )
    if atomic:
        layers["conv_to_output_hidden"] = (
            rnn.NodewiseLinear,
            {"irreps_out": CartesianTensor(formula=hparams["output_formula"].lower()), "out_field": OUT_FIELD_NAME},
        )
        return create_sequential_module(OrderedDict(layers))
    layers["conv_to_output_hidden"] = (
        rnn.NodewiseLinear,
        {"irreps_out": hparams["conv_to_output_hidden_irreps_out"], "out_field": OUT_FIELD_NAME},
    )
    layers["output_pooling"] = (
        rnn.NodewiseReduce,
        {"field": OUT_FIELD_NAME, "out_field": OUT_FIELD_NAME, "reduce": hparams["reduce"]},
    )
    return create_sequential_module(OrderedDict(layers))


class ToCartesian(torch.nn.Module):
    def __init__(self, formula):
        super().__init__()
        self.ct = CartesianTensor(formula)

    def forward(self, data):
        return self.ct.to_cartesian(data)


class ScalarTensorOracle(torch.nn.Module):
    """backbone + extra_layers_dict['out_layer'] (+ to_cartesian) with the reference's parameter names."""

    def __init__(self, backbone_hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]):
        super().__init__()
        self.backbone = create_model(backbone_hparams, dataset_hparams)
        formula = backbone_hparams["output_formula"].lower()
        irreps_out = o3.Irreps("0e") if formula == "scalar" else CartesianTensor(formula=formula)
        irreps_in = backbone_hparams["conv_to_output_hidden_irreps_out"]
        self.extra_layers_dict = torch.nn.ModuleDict({"out_layer": o3.Linear(irreps_in=irreps_in, irreps_out=irreps_out)})
        if backbone_hparams["output_format"] == "cartesian" and formula != "scalar":
            self.to_cartesian = ToCartesian(formula)
        else:
            self.to_cartesian = None

    def decode(self, model_input: Dict[str, torch.Tensor]) -> torch.Tensor:
        out = self.backbone(model_input)[OUT_FIELD_NAME]
        out = self.extra_layers_dict["out_layer"](out)
        if self.to_cartesian is not None:
            out = self.to_cartesian(out)
        return out

    forward = decode


class AtomicTensorOracle(torch.nn.Module):
    """AtomicTensorModel (model_factory/tfn_atomic_tensor.py:31-77): one tensor per atom, no out_layer."""

    def __init__(self, backbone_hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]):
        super().__init__()
        self.backbone = create_model(backbone_hparams, dataset_hparams, atomic=True)
        formula = backbone_hparams["output_formula"].lower()
        if backbone_hparams["output_format"] == "cartesian" and formula != "scalar":
            self.to_cartesian = ToCartesian(formula)
        else:
            self.to_cartesian = None

    def decode(self, model_input: Dict[str, torch.Tensor]) -> torch.Tensor:
        out = self.backbone(model_input)[OUT_FIELD_NAME]
        if self.to_cartesian is not None:
            out = self.to_cartesian(out)
        return out

    forward = decode
