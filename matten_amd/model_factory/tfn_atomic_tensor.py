"""
Per-atom tensor model (e.g. NMR shielding tensors, ``ij=ji``): the conv stack of the scalar/tensor model with a node
head that maps straight to the Cartesian-tensor irreps and no pooling.  Mirrors the reference factory
(model_factory/tfn_atomic_tensor.py:31-199): ``AtomicTensorModel(...).forward(batch, mode, task_name)``,
``decode``, ``backbone``; there is no ``out_layer``.  ``select_atoms`` is the ``atom_selector`` step of the
reference's ``shared_step`` (model/model.py:340-345).
"""
from typing import Any, Dict, Tuple

import torch
from torch import Tensor

from ..nn.nodewise import NodewiseLinear
from ..utils import CartesianTensor, ToCartesian
from .tfn_scalar_tensor import OUT_FIELD_NAME, ScalarTensorModel
from .tfn_scalar_tensor import create_model as _create_scalar_tensor_model
from .utils import create_sequential_module


def create_model(hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]):
    """Embeddings + convs as in tfn_scalar_tensor.create_model; head = NodewiseLinear -> CartesianTensor(formula)."""
    formula = hparams["output_formula"].lower()
    return _create_scalar_tensor_model(
        dict(hparams, conv_to_output_hidden_irreps_out=CartesianTensor(formula=formula)), dataset_hparams, pooling=False
    )


class AtomicTensorModel(ScalarTensorModel):
    # --- reference: AtomicTensorModel.init_backbone, tfn_atomic_tensor.py:32-61 ---------------------
    def init_backbone(self, backbone_hparams, dataset_hparams=None) -> Tuple[torch.nn.Module, Dict]:
        backbone = create_model(backbone_hparams, dataset_hparams)
        formula = backbone_hparams["output_formula"].lower()
        if backbone_hparams["output_format"] == "cartesian" and formula != "scalar":
            self.to_cartesian = ToCartesian(formula)
        else:
            self.to_cartesian = None
        return backbone, {}

    # --- reference: AtomicTensorModel.decode, tfn_atomic_tensor.py:63-79 ------------------------------
    def decode(self, model_input) -> Dict[str, Tensor]:
        out = self.backbone(model_input)[OUT_FIELD_NAME]
        if self.to_cartesian is not None:
            out = self.to_cartesian(out)
        names = list(self.tasks.keys())
        assert len(names) == 1, f"only works for 1 target, get{len(names)}"
        return {names[0]: out}

    def preprocess_batch(self, batch):
        graphs, labels = super().preprocess_batch(batch)
        src = getattr(batch, "y", None) if hasattr(batch, "tensor_property_to_dict") else graphs
        if src is not None and "atom_selector" in src:
            labels["atom_selector"] = src["atom_selector"]
        return graphs, labels

    @staticmethod
    def select_atoms(preds: Dict[str, Tensor], labels: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """keep the predictions of the atoms that carry a target (model/model.py:340-345)"""
        if "atom_selector" in labels:
            sel = labels["atom_selector"].to(torch.bool)
            return {k: v[sel] for k, v in preds.items()}
        return preds
