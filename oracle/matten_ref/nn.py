"""
CPU restatement of the reference's conv-stack modules (``src/matten/nn`` of wengroup/matten)
on top of ``oracle.e3nn_lite``.  ORACLE / TEST INFRASTRUCTURE (see oracle/__init__.py).

Same class names, constructor arguments, attribute names (=> same ``state_dict`` keys) and the
same ``forward(data: Dict[str, Tensor]) -> Dict[str, Tensor]`` convention as the reference, so
a state_dict moves between this oracle and the HIP product unchanged.  Every class cites the
reference lines it follows.  Only the branches the two model factories reach are restated.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as fn
from torch import Tensor

from ..e3nn_lite import o3
from ..e3nn_lite.math import soft_one_hot_linspace
from ..e3nn_lite.nn import BatchNorm, FullyConnectedNet, Gate, NormActivation
from ..e3nn_lite.o3 import FullyConnectedTensorProduct, Irrep, Irreps, TensorProduct
from ..e3nn_lite.scatter import scatter


class DataKey:
    """data/_key.py:14-49"""

    POSITIONS = "pos"
    NODE_ATTRS = "node_attrs"
    NODE_FEATURES = "node_features"
    EDGE_INDEX = "edge_index"
    EDGE_CELL_SHIFT = "edge_cell_shift"
    EDGE_VECTORS = "edge_vectors"
    EDGE_LENGTH = "edge_lengths"
    EDGE_ATTRS = "edge_attrs"
    EDGE_EMBEDDING = "edge_embedding"
    CELL = "cell"
    NUM_NEIGH = "num_neigh"
    ATOMIC_NUMBERS = "atomic_numbers"
    SPECIES_INDEX = "species_index"
    BATCH = "batch"


class ShiftedSoftPlus(torch.nn.Module):
    """nn/_nequip.py:17-39"""

    def __init__(self, beta=1, threshold=20):
        super().__init__()
        self.softplus = torch.nn.Softplus(beta=beta, threshold=threshold)
        self._log2 = 0.6931471805599453

    def forward(self, x):
        return self.softplus(x) - self._log2


# nn/utils.py:14-26
ACTIVATION = {
    "e": {"ssp": ShiftedSoftPlus(), "silu": fn.silu, "sigmoid": torch.sigmoid},
    "o": {"abs": torch.abs, "tanh": torch.tanh},
}


# ---------------------------------------------------------------------------------------
# irreps bookkeeping (data/irreps.py:17-165), reduced to what the factories use
# ---------------------------------------------------------------------------------------
def _fix_irreps_dict(d):
    return {k: (v if v is None else Irreps(v)) for k, v in d.items()}


class ModuleIrreps:
    def init_irreps(self, irreps_in=None, irreps_out=None, required_keys_irreps_in=None):
        irreps_in = {} if irreps_in is None else irreps_in
        irreps_in = _fix_irreps_dict(irreps_in)
        irreps_in = dict(irreps_in)
        irreps_in[DataKey.POSITIONS] = Irreps("1o")  # data/irreps.py:146-151
        irreps_in[DataKey.EDGE_INDEX] = None  # data/irreps.py:153-160
        irreps_out = _fix_irreps_dict(irreps_out or {})
        for k in required_keys_irreps_in or []:
            if k not in irreps_in:
                raise ValueError(f"This module {type(self)} requires `{k}` in `irreps_in`.")
        self._irreps_in = irreps_in
        self._irreps_out = irreps_in.copy()
        self._irreps_out.update(irreps_out)

    @property
    def irreps_in(self):
        return self._irreps_in

    @property
    def irreps_out(self):
        return self._irreps_out


def _check_irreps_compatible(ir1, ir2):
    return all(ir1[k] == ir2[k] for k in ir1 if k in ir2)


class Sequential(torch.nn.Sequential, ModuleIrreps):
    """nn/sequential.py:9-48"""

    def __init__(self, module_dict: "OrderedDict[str, torch.nn.Module]"):
        module_list = list(module_dict.values())
        for i, (m1, m2) in enumerate(zip(module_list, module_list[1:])):
            if not _check_irreps_compatible(m1.irreps_out, m2.irreps_in):
                raise ValueError(f"Output irreps of module {i} incompatible with input irreps of module {i + 1}")
        self.init_irreps(irreps_in=module_list[0].irreps_in, irreps_out=module_list[-1].irreps_out)
        torch.nn.Sequential.__init__(self, module_dict)


# ---------------------------------------------------------------------------------------
# embedding (nn/embedding.py)
# ---------------------------------------------------------------------------------------
class _AtomicNumberToIndex(torch.nn.Module):
    """nn/embedding.py:206-263"""

    def __init__(self, allowed_atomic_numbers: List[int]):
        super().__init__()
        allowed = torch.as_tensor(sorted(allowed_atomic_numbers), dtype=torch.long)
        num_species = len(allowed)
        self.register_buffer("_min_Z", allowed.min())
        self.register_buffer("_max_Z", allowed.max())
        self.register_buffer("_num_species", torch.as_tensor(num_species))
        Z_to_index = torch.full((1 + self._max_Z - self._min_Z,), -1, dtype=torch.long)
        Z_to_index[allowed - self._min_Z] = torch.arange(num_species).to(torch.long)
        self.register_buffer("_Z_to_index", Z_to_index)

    def forward(self, atomic_numbers: Tensor) -> Tensor:
        if atomic_numbers.min() < self._min_Z or atomic_numbers.max() > self._max_Z:
            raise RuntimeError(
                "Invalid atomic numbers. Expect atomic numbers to be in the range "
                f"[{self._min_Z}, {self._max_Z}], but got min {atomic_numbers.min()} "
                f"and max {atomic_numbers.max()}"
            )
        index = self._Z_to_index[atomic_numbers - self._min_Z]
        if index.min() < 0:
            supported = [Z + int(self._min_Z) for Z, idx in enumerate(self._Z_to_index) if idx != -1]
            for i, val in enumerate(index):
                if val == -1:
                    raise RuntimeError(
                        f"Expect atomic numbers to be in {supported}, "
                        f"got invalid atomic numbers `{atomic_numbers[i]}` for data point `{i}`."
                    )
        return index

    @property
    def num_species(self):
        return self._num_species


class SpeciesEmbedding(ModuleIrreps, torch.nn.Module):
    """nn/embedding.py:12-110 (use_atom_feats: :59-68 feature width, :103-105 hstack with data["atom_feats"])"""

    def __init__(self, irreps_in=None, embedding_dim: int = 16, allowed_species: List[int] = None,
                 use_atom_feats: bool = False, atom_feats_dim: int = None, **_ignored):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.use_atom_feats = use_atom_feats
        self.atomic_number_to_index = _AtomicNumberToIndex(allowed_species)
        self.num_species = int(self.atomic_number_to_index.num_species)
        if use_atom_feats and atom_feats_dim is None:
            raise ValueError("`atom_feats_dim` must be provided if `use_atom_feats` is True.")
        feats_dim = embedding_dim + (atom_feats_dim if use_atom_feats else 0)
        irreps_out = {
            DataKey.NODE_ATTRS: Irreps(f"{self.num_species}x0e"),
            DataKey.NODE_FEATURES: Irreps(f"{feats_dim}x0e"),
        }
        self.init_irreps(irreps_in, irreps_out)
        self.linear = torch.nn.Linear(self.num_species, embedding_dim)

    def forward(self, data):
        if DataKey.SPECIES_INDEX in data:
            type_numbers = data[DataKey.SPECIES_INDEX]
        elif DataKey.ATOMIC_NUMBERS in data:
            type_numbers = self.atomic_number_to_index(data[DataKey.ATOMIC_NUMBERS])
            data[DataKey.SPECIES_INDEX] = type_numbers
        else:
            raise ValueError("Nothing in `data` to encode. Need either species_index or atomic_numbers")
        attrs = torch.nn.functional.one_hot(type_numbers, num_classes=self.num_species).to(self.linear.weight.dtype)
        embed = self.linear(attrs)
        if self.use_atom_feats:
            embed = torch.hstack((embed, data["atom_feats"].to(embed.dtype)))
        data[DataKey.NODE_ATTRS] = attrs
        data[DataKey.NODE_FEATURES] = embed
        return data


def with_edge_vectors(data, with_lengths: bool = True):
    """nn/_nequip.py:214-268"""
    if DataKey.EDGE_VECTORS in data:
        if with_lengths and DataKey.EDGE_LENGTH not in data:
            data[DataKey.EDGE_LENGTH] = torch.linalg.norm(data[DataKey.EDGE_VECTORS], dim=-1)
        return data
    pos = data[DataKey.POSITIONS]
    edge_index = data[DataKey.EDGE_INDEX]
    edge_vec = pos[edge_index[1]] - pos[edge_index[0]]
    if DataKey.CELL in data:
        cell = data[DataKey.CELL].view(-1, 3, 3)
        edge_cell_shift = data[DataKey.EDGE_CELL_SHIFT]
        if cell.shape[0] > 1:
            batch = data[DataKey.BATCH]
            edge_vec = edge_vec + torch.einsum("ni,nij->nj", edge_cell_shift, cell[batch[edge_index[0]]])
        else:
            edge_vec = edge_vec + torch.einsum("ni,ij->nj", edge_cell_shift, cell.squeeze(0))
    data[DataKey.EDGE_VECTORS] = edge_vec
    if with_lengths:
        data[DataKey.EDGE_LENGTH] = torch.linalg.norm(edge_vec, dim=-1)
    return data


def with_batch(data):
    """nn/_nequip.py:272-285"""
    if DataKey.BATCH in data:
        return data
    pos = data[DataKey.POSITIONS]
    data[DataKey.BATCH] = torch.zeros(len(pos), dtype=torch.long, device=pos.device)
    return data


class SphericalHarmonicEdgeAttrs(ModuleIrreps, torch.nn.Module):
    """nn/_nequip.py:130-176"""

    def __init__(self, irreps_edge_sh, edge_sh_normalization="component", edge_sh_normalize=True, irreps_in=None,
                 out_field: str = DataKey.EDGE_ATTRS):
        super().__init__()
        self.out_field = out_field
        if isinstance(irreps_edge_sh, int):
            self.irreps_edge_sh = Irreps.spherical_harmonics(irreps_edge_sh)
        else:
            self.irreps_edge_sh = Irreps(irreps_edge_sh)
        self.init_irreps(irreps_in=irreps_in, irreps_out={out_field: self.irreps_edge_sh})
        self.sh = o3.SphericalHarmonics(self.irreps_edge_sh, edge_sh_normalize, edge_sh_normalization)

    def forward(self, data):
        data = with_edge_vectors(data, with_lengths=False)
        data[self.out_field] = self.sh(data[DataKey.EDGE_VECTORS])
        return data


class EdgeLengthEmbedding(ModuleIrreps, torch.nn.Module):
    """nn/embedding.py:158-203"""

    def __init__(self, irreps_in=None, out_field=DataKey.EDGE_EMBEDDING, num_basis=10, start=0.0, end=5.0,
                 basis="bessel", cutoff=True):
        super().__init__()
        self.num_basis, self.start, self.end, self.basis, self.cutoff = num_basis, start, end, basis, cutoff
        self.out_field = out_field
        self.init_irreps(irreps_in, irreps_out={out_field: Irreps(f"{num_basis}x0e")})

    def forward(self, data):
        data = with_edge_vectors(data, with_lengths=True)
        emb = soft_one_hot_linspace(
            data[DataKey.EDGE_LENGTH], start=self.start, end=self.end, number=self.num_basis, basis=self.basis,
            cutoff=self.cutoff,
        )
        data[self.out_field] = emb.mul(self.num_basis**0.5)
        return data


# ---------------------------------------------------------------------------------------
# nn/utils.py
# ---------------------------------------------------------------------------------------
def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """nn/utils.py:358-367"""
    irreps_in1 = Irreps(irreps_in1).simplify()
    irreps_in2 = Irreps(irreps_in2).simplify()
    ir_out = Irrep(ir_out)
    for _, ir1 in irreps_in1:
        for _, ir2 in irreps_in2:
            if ir_out in ir1 * ir2:
                return True
    return False


class ActivationLayer(torch.nn.Module):
    """nn/utils.py:29-167"""

    def __init__(self, tp_irreps_in1, tp_irreps_in2, tp_irreps_out, *, activation_type="gate",
                 activation_scalars: Dict[str, str] = None, activation_gates: Dict[str, str] = None):
        super().__init__()
        key_mapping = {"e": 1, "o": -1}
        if activation_scalars is None:
            activation_scalars = {1: ACTIVATION["e"]["ssp"], -1: ACTIVATION["o"]["tanh"]}
        else:
            activation_scalars = {key_mapping[k]: ACTIVATION[k][v] for k, v in activation_scalars.items()}
        if activation_gates is None:
            activation_gates = {1: ACTIVATION["e"]["ssp"], -1: ACTIVATION["o"]["abs"]}
        else:
            activation_gates = {key_mapping[k]: ACTIVATION[k][v] for k, v in activation_gates.items()}

        ir_tmp, _, _ = Irreps(tp_irreps_out).sort()
        tp_irreps_out = ir_tmp.simplify()

        irreps_scalars = Irreps(
            [(mul, ir) for mul, ir in tp_irreps_out if ir.l == 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
        )
        irreps_gated = Irreps(
            [(mul, ir) for mul, ir in tp_irreps_out if ir.l > 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
        )
        if activation_type == "norm":   # nn/utils.py:142-150
            self.activation = NormActivation(
                irreps_in=(irreps_scalars + irreps_gated).simplify(),
                scalar_nonlinearity=activation_scalars[1],   # "norm is an even scalar, so activation_scalars[1]"
                normalize=True, epsilon=1e-8, bias=False,
            )
            return
        if activation_type != "gate":
            raise ValueError(f"Support `activation_type` includes ('gate', 'norm'), got {activation_type}")
        if irreps_gated.dim > 0:
            if tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0e"):
                ir = "0e"
            elif tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0o"):
                ir = "0o"
            else:
                raise ValueError("unable to produce gates")
        else:
            ir = None
        irreps_gates = Irreps([(mul, ir) for mul, _ in irreps_gated]).simplify()
        self.activation = Gate(
            irreps_scalars=irreps_scalars,
            act_scalars=[activation_scalars[ir.p] for _, ir in irreps_scalars],
            irreps_gates=irreps_gates,
            act_gates=[activation_gates[ir.p] for _, ir in irreps_gates],
            irreps_gated=irreps_gated,
        )

    def forward(self, x):
        return self.activation(x)

    @property
    def irreps_in(self):
        return self.activation.irreps_in

    @property
    def irreps_out(self):
        return self.activation.irreps_out


class UVUTensorProduct(torch.nn.Module):
    """nn/utils.py:170-277"""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, *, internal_and_share_weights: bool = False,
                 mlp_input_size: int = None, mlp_hidden_size: int = 8, mlp_num_hidden_layers: int = 1,
                 mlp_activation: Callable = ACTIVATION["e"]["ssp"]):
        super().__init__()
        irreps_in1, irreps_in2, irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        irreps_mid = []
        instructions = []
        for i, (mul, ir_in1) in enumerate(irreps_in1):
            for j, (_, ir_in2) in enumerate(irreps_in2):
                for ir_out in ir_in1 * ir_in2:
                    # nn/utils.py:210 -- the second clause compares an Irrep to an Irreps and is never true
                    if ir_out in irreps_out or ir_out == Irreps("0e"):
                        k = len(irreps_mid)
                        irreps_mid.append((mul, ir_out))
                        instructions.append((i, j, k, "uvu", True))
        irreps_mid = Irreps(irreps_mid)
        assert irreps_mid.dim > 0
        self.irreps_mid, permutation, _ = irreps_mid.sort()
        instructions = [(i_1, i_2, permutation[i_out], mode, train) for i_1, i_2, i_out, mode, train in instructions]
        self.tp = TensorProduct(
            irreps_in1, irreps_in2, self.irreps_mid, instructions,
            internal_weights=internal_and_share_weights, shared_weights=internal_and_share_weights,
        )
        if not internal_and_share_weights:
            assert mlp_input_size is not None
            layer_sizes = [mlp_input_size] + mlp_num_hidden_layers * [mlp_hidden_size] + [self.tp.weight_numel]
            self.weight_nn = FullyConnectedNet(layer_sizes, act=mlp_activation)
        else:
            self.weight_nn = None

    def forward(self, data1, data2, data_weight=None):
        weight = self.weight_nn(data_weight) if self.weight_nn is not None else None
        return self.tp(data1, data2, weight)

    @property
    def irreps_out(self):
        return self.irreps_mid.simplify()


def _segment_mean(x, batch, n_seg):
    """nn/utils.py:591-618 global_mean_pool: per-graph mean over the node dimension"""
    out = torch.zeros((n_seg,) + tuple(x.shape[1:]), dtype=x.dtype).index_add_(0, batch, x)
    count = torch.zeros(n_seg, dtype=x.dtype).index_add_(0, batch, torch.ones(batch.shape[0], dtype=x.dtype))
    return out / count.clamp(min=1).reshape((n_seg,) + (1,) * (x.dim() - 1))


class InstanceNorm(torch.nn.Module):
    """nn/utils.py:448-588 (the reference's own graph-wise InstanceNorm; affine=True, reduce='mean',
    normalization='component', the defaults NormalizationLayer uses)"""

    def __init__(self, irreps, eps=1e-5):
        super().__init__()
        self.irreps = Irreps(irreps)
        self.eps = eps
        num_scalar = sum(mul for mul, ir in self.irreps if ir.l == 0)
        self.weight = torch.nn.Parameter(torch.ones(self.irreps.num_irreps))
        self.bias = torch.nn.Parameter(torch.zeros(num_scalar))

    def forward(self, input, batch):
        n_seg = int(batch.max()) + 1 if batch.numel() else 1
        fields, ix, iw, ib = [], 0, 0, 0
        for mul, ir in self.irreps:
            d = ir.dim
            field = input[:, ix: ix + mul * d].reshape(-1, mul, d)
            ix += mul * d
            if ir.l == 0:                                                       # :531-538
                field = field - _segment_mean(field, batch, n_seg).reshape(-1, mul, 1)[batch]
            field_norm = field.pow(2).mean(-1)                                  # 'component', :545
            field_norm = _segment_mean(field_norm, batch, n_seg)                # reduce 'mean', :552
            field_norm = (field_norm + self.eps).pow(-0.5)                      # :560
            field_norm = field_norm * self.weight[None, iw: iw + mul]           # :562-565
            iw += mul
            field = field * field_norm[batch].reshape(-1, mul, 1)               # :567-569
            if d == 1:                                                          # :571-574
                field = field + self.bias[ib: ib + mul].reshape(mul, 1)
                ib += mul
            fields.append(field.reshape(-1, mul * d))
        assert ix == input.shape[-1]
        return torch.cat(fields, dim=-1)


class NormalizationLayer(torch.nn.Module):
    """nn/utils.py:397-437"""

    def __init__(self, irreps, method: str = None):
        super().__init__()
        self.method = method
        assert method in ("batch", "instance", "none", None), f"Unsupported normalization {method}"
        self.n = BatchNorm(irreps) if method == "batch" else InstanceNorm(irreps) if method == "instance" else None

    def forward(self, x, batch):
        if self.method == "batch":
            x = self.n(x)
        elif self.method == "instance":
            x = self.n(x, batch)
        return x


# ---------------------------------------------------------------------------------------
# nn/conv.py
# ---------------------------------------------------------------------------------------
class PointConv(ModuleIrreps, torch.nn.Module):
    """nn/conv.py:26-143"""

    def __init__(self, irreps_in, conv_layer_irreps, fc_num_hidden_layers: int = 1, fc_hidden_size: int = 8,
                 avg_num_neighbors=None):
        super().__init__()
        self.avg_num_neighbors = avg_num_neighbors
        self.init_irreps(irreps_in)
        node_feats_irreps_in = self.irreps_in[DataKey.NODE_FEATURES]
        node_attrs_irreps = self.irreps_in[DataKey.NODE_ATTRS]
        edge_attrs_irreps = self.irreps_in[DataKey.EDGE_ATTRS]
        conv_layer_irreps = Irreps(conv_layer_irreps)

        self.lin1 = FullyConnectedTensorProduct(node_feats_irreps_in, node_attrs_irreps, node_feats_irreps_in)
        self.tp = UVUTensorProduct(
            node_feats_irreps_in, edge_attrs_irreps, conv_layer_irreps,
            mlp_input_size=self.irreps_in[DataKey.EDGE_EMBEDDING].dim,
            mlp_hidden_size=fc_hidden_size, mlp_num_hidden_layers=fc_num_hidden_layers,
            mlp_activation=ACTIVATION["e"]["silu"],
        )
        tp_irreps_out = self.tp.irreps_out
        self.lin2 = FullyConnectedTensorProduct(tp_irreps_out, node_attrs_irreps, conv_layer_irreps)
        self.sc = FullyConnectedTensorProduct(node_feats_irreps_in, node_attrs_irreps, conv_layer_irreps)
        self.irreps_out[DataKey.NODE_FEATURES] = conv_layer_irreps

    def forward(self, data):
        node_feats = data[DataKey.NODE_FEATURES]
        node_attrs = data[DataKey.NODE_ATTRS]
        edge_attrs = data[DataKey.EDGE_ATTRS]
        edge_embedding = data[DataKey.EDGE_EMBEDDING]
        edge_src, edge_dst = data[DataKey.EDGE_INDEX]

        node_self_connection = self.sc(node_feats, node_attrs)
        node_feats = self.lin1(node_feats, node_attrs)
        msg = self.tp(node_feats[edge_src], edge_attrs, edge_embedding)
        aggregated_msg = scatter(msg, edge_dst, dim_size=len(node_feats), dim=0)
        if self.avg_num_neighbors is not None:
            aggregated_msg = aggregated_msg.div(self.avg_num_neighbors**0.5)
        else:
            num_neigh = data[DataKey.NUM_NEIGH].reshape(-1, 1)
            aggregated_msg = aggregated_msg.div(num_neigh**0.5)
        node_conv_out = self.lin2(aggregated_msg, node_attrs)
        data[DataKey.NODE_FEATURES] = node_self_connection + node_conv_out
        return data


class PointConvWithActivation(ModuleIrreps, torch.nn.Module):
    """nn/conv.py:146-215"""

    def __init__(self, irreps_in, conv_layer_irreps, fc_num_hidden_layers: int = 1, fc_hidden_size: int = 8,
                 avg_num_neighbors=None, activation_type: str = "gate",
                 activation_scalars: Dict[str, str] = {"e": "silu", "o": "tanh"},
                 activation_gates: Dict[str, str] = {"e": "sigmoid", "o": "tanh"}, normalization: str = None):
        super().__init__()
        self.init_irreps(irreps_in)
        node_feats_irreps_in = self.irreps_in[DataKey.NODE_FEATURES]
        edge_attrs_irreps = self.irreps_in[DataKey.EDGE_ATTRS]
        conv_layer_irreps = Irreps(conv_layer_irreps)
        self.act = ActivationLayer(
            node_feats_irreps_in, edge_attrs_irreps, conv_layer_irreps, activation_type=activation_type,
            activation_scalars=activation_scalars, activation_gates=activation_gates,
        )
        self.conv = PointConv(
            irreps_in=self.irreps_in, conv_layer_irreps=self.act.irreps_in,
            fc_num_hidden_layers=fc_num_hidden_layers, fc_hidden_size=fc_hidden_size,
            avg_num_neighbors=avg_num_neighbors,
        )
        self.norm = NormalizationLayer(self.act.irreps_out, method=normalization)
        self.irreps_out[DataKey.NODE_FEATURES] = self.act.irreps_out

    def forward(self, data):
        batch = data[DataKey.BATCH]
        data = self.conv(data)
        x = data[DataKey.NODE_FEATURES]
        x = self.act(x)
        x = self.norm(x, batch)
        data[DataKey.NODE_FEATURES] = x
        return data


# ---------------------------------------------------------------------------------------
# nn/nodewise.py
# ---------------------------------------------------------------------------------------
class NodewiseLinear(ModuleIrreps, torch.nn.Module):
    """nn/nodewise.py:89-117"""

    def __init__(self, irreps_in, irreps_out=None, field: str = DataKey.NODE_FEATURES, out_field: Optional[str] = None):
        super().__init__()
        self.field = field
        self.out_field = out_field if out_field is not None else field
        if irreps_out is None:
            irreps_out = irreps_in[self.field]
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: irreps_out},
                         required_keys_irreps_in=[self.field])
        self.linear = o3.Linear(irreps_in=self.irreps_in[field], irreps_out=self.irreps_out[self.out_field])

    def forward(self, data):
        data[self.out_field] = self.linear(data[self.field])
        return data


class NodewiseReduce(ModuleIrreps, torch.nn.Module):
    """nn/nodewise.py:120-148"""

    def __init__(self, irreps_in, field: str, out_field: Optional[str] = None, reduce: str = "sum"):
        super().__init__()
        assert reduce in ("sum", "mean", "min", "max")
        self.reduce = reduce
        self.field = field
        self.out_field = f"{reduce}_{field}" if out_field is None else out_field
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: irreps_in[self.field]},
                         required_keys_irreps_in=[self.field])

    def forward(self, data):
        with_batch(data)
        data[self.out_field] = scatter(data[self.field], data[DataKey.BATCH], dim=0, reduce=self.reduce)
        return data
