"""
``TensorDataModule``: the data module the reference's training script builds (scripts/train_materials_tensor.py:35-37;
dataset/structure_scalar_tensor.py:368-666), reduced to what feeds the message-passing path: the reference's JSON files
(``structure`` column = pymatgen ``Structure.as_dict()``, target column = Cartesian tensor) -> crystal graphs with the
neighbour list of ``matten_amd.data.graph`` -> collated batch dicts whose target is the irreps vector of the tensor
(``CartesianTensor.from_cartesian``, reference :262-267).  pymatgen / pandas / PyG / Lightning are not needed.

Not carried over (outside the path; ``NotImplementedError`` when asked for): featurizers (``global_featurizer``,
``atom_featurizer``), scalar targets, on-disk caching (``reuse``); target normalisation is available through
``normalize_tensor_target`` + ``matten_amd.data.transform``.
"""
import os
from typing import Any, Dict, Iterator, List, Optional

import numpy as np
import torch

from ..data.graph import collate, crystal_graph
from ..data.io import structures_from_json
from ..utils import CartesianTensorWrapper


class _Loader:
    """the subset of torch_geometric's DataLoader the training loop uses: iteration over collated batches"""

    def __init__(self, graphs: List[Dict[str, torch.Tensor]], batch_size: int = 1, shuffle: bool = False,
                 device=None, seed: int = 0, **ignored):
        self.graphs, self.batch_size, self.shuffle, self.device = graphs, int(batch_size), bool(shuffle), device
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self) -> int:
        return -(-len(self.graphs) // self.batch_size)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = len(self.graphs)
        order = torch.randperm(n, generator=self._gen).tolist() if self.shuffle else list(range(n))
        for lo in range(0, n, self.batch_size):
            yield collate([self.graphs[i] for i in order[lo:lo + self.batch_size]], device=self.device)


class TensorDataModule:
    def __init__(
        self,
        trainset_filename: str,
        valset_filename: str,
        testset_filename: str,
        *,
        r_cut: float,
        tensor_target_name: str,
        tensor_target_format: str = "irreps",
        tensor_target_formula: str = "ijkl=jikl=klij",
        tensor_target_scale: float = 1.0,
        normalize_tensor_target: bool = False,
        root: str = ".",
        reuse: bool = True,
        loader_kwargs: Optional[Dict[str, Any]] = None,
        device=None,
        **unsupported,
    ):
        bad = {k: v for k, v in unsupported.items() if v not in (None, False, [], {})
               and k in ("global_featurizer", "atom_featurizer", "scalar_target_names", "tensor_target_weight", "atom_selector")}
        if bad:
            raise NotImplementedError(f"TensorDataModule options outside the accelerated path: {sorted(bad)}")
        if normalize_tensor_target:
            raise NotImplementedError("normalize_tensor_target: standardise with matten_amd.data.transform.TensorTargetTransform")
        self.files = {"train": trainset_filename, "val": valset_filename, "test": testset_filename}
        self.root, self.r_cut = root, float(r_cut)
        self.tensor_target_name, self.tensor_target_format = tensor_target_name, tensor_target_format
        self.tensor_target_formula, self.tensor_target_scale = tensor_target_formula, float(tensor_target_scale)
        self.loader_kwargs = dict(loader_kwargs or {})
        self.device = device
        self._data: Dict[str, List[Dict[str, torch.Tensor]]] = {}

    # Lightning's DataModule protocol
    def prepare_data(self):
        pass

    def _load(self, filename: str) -> List[Dict[str, torch.Tensor]]:
        path = filename if os.path.isabs(filename) else os.path.join(self.root, filename)
        rows = structures_from_json(path, target_columns=(self.tensor_target_name,))
        converter = CartesianTensorWrapper(self.tensor_target_formula)
        graphs = []
        for r in rows:
            y = {}
            if self.tensor_target_name in r:
                t = torch.as_tensor(r[self.tensor_target_name], dtype=torch.float64) * self.tensor_target_scale
                if self.tensor_target_format == "irreps":   # reference dataset/structure_scalar_tensor.py:262-267
                    t = converter.from_cartesian(t)
                y[self.tensor_target_name] = t.to(torch.float32).unsqueeze(0)
            graphs.append(crystal_graph(r["cart_coords"], r["lattice"], r["atomic_numbers"], self.r_cut, y=y))
        return graphs

    def setup(self, stage: Optional[str] = None):
        cache: Dict[str, List] = {}
        for mode, fn in self.files.items():
            if fn not in cache:
                cache[fn] = self._load(fn)
            self._data[mode] = cache[fn]

    @property
    def train_data(self):
        return self._data["train"]

    def train_dataloader(self):
        return _Loader(self._data["train"], device=self.device, **self.loader_kwargs)

    def val_dataloader(self):
        return _Loader(self._data["val"], device=self.device, **dict(self.loader_kwargs, shuffle=False))

    def test_dataloader(self):
        return _Loader(self._data["test"], device=self.device, **dict(self.loader_kwargs, shuffle=False))

    # reference dataset/structure_scalar_tensor.py:640-666
    def get_to_model_info(self) -> Dict[str, Any]:
        z = set()
        num_neigh = []
        for g in self._data["train"]:
            z.update(g["atomic_numbers"].tolist())
            num_neigh.append(g["num_neigh"])
        return {
            "allowed_species": tuple(sorted(z)),
            "average_num_neighbors": torch.mean(torch.cat(num_neigh)).item(),
            "global_feats_size": None,
            "atom_feats_size": None,
        }
