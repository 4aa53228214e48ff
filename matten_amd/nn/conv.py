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
from ._tables import DerivedWeight, DeviceTables, WeightSlice
from .utils import ActivationLayer, NormalizationLayer, SpeciesLinear, UVUTensorProduct


import os as _os

GATE_APPLIED = "_amd_gate_applied"    # batch-dict marker: the conv layer's lin2 kernel already applied Gate (+ BatchNorm)
# Gate + eval BatchNorm inside the lin2 kernel's epilogue (matten_agg_linear_gate); MATTEN_GATE_FUSE=0 keeps them separate
GATE_FUSE = _os.environ.get("MATTEN_GATE_FUSE", "1") != "0"
KEPT_ONLY = "_amd_kept_irreps_only"   # batch-dict marker: node_features hold the irreps of the inference view only
# inference: skip the output irreps of the last conv layer nothing reads (PointConv.build_inference_view); 0 = run them
DEAD_PATH_ELIMINATION = _os.environ.get("MATTEN_DEAD_PATH_ELIMINATION", "1") != "0"
AGG_KM_MIN_ROWS = int(_os.environ.get("MATTEN_AGG_KM_MIN_ROWS", "8192"))   # nodes per batch from which lin2 streams component-major rows


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
        self.__dict__["_ctor"] = dict(irreps_in=dict(irreps_in), fc_num_hidden_layers=fc_num_hidden_layers,
                                      fc_hidden_size=fc_hidden_size, avg_num_neighbors=avg_num_neighbors)
        self.__dict__["_view"] = None   # inference view with fewer output irreps (build_inference_view)
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
        from .. import plan as _plan
        # component-major neighbour sums + streaming lin2 (matten_agg_linear): the inference path of the two-kernel conv
        import os
        self.agg_plan = None
        if self.tp.impl == "fused" and os.environ.get("MATTEN_AGG_LAYOUT", "km") == "km":
            ap = _plan.plan_agg_linear(self.tp.plan, n_species, conv_layer_irreps)
            if ap is not None and self._agg_fits(ap):
                self.agg_plan = ap
                self._agg_tables = DeviceTables(entries=ap.entries, io=ap.io_table, blocks=ap.blocks, gather=ap.gather,
                                                scale=ap.scale)
                self._agg_wtab = DerivedWeight(self._pack_agg_weights)

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

    @staticmethod
    def _agg_fits(ap) -> bool:
        """the layer's tables fit the LDS matten_agg_linear may take (wider layers keep the mul_ir path)"""
        from .. import _lib
        try:
            lib = _lib.load()
        except Exception:  # noqa: BLE001  (no library: construction-time planning on a build box)
            return 4 * ap.w_stride <= 40 * 1024
        return lib.matten_agg_linear_lds_bytes(ap.w_stride, len(ap.io_table), len(ap.blocks)) <= lib.matten_agg_linear_max_lds_bytes()

    def _pack_agg_weights(self, w: torch.Tensor) -> torch.Tensor:
        """lin2.weight (flat, reference layout) -> [S, w_stride] MFMA A fragments of matten_agg_linear"""
        t, dev = self._agg_tables, w.device
        gather, scale = t.get("gather", dev), t.get("scale", dev)
        return torch.where(gather >= 0, w[gather.clamp(min=0)] * scale[None, :], w.new_zeros(())).contiguous()

    # ---- dead-output elimination (inference) ------------------------------------------------------------------------
    def build_inference_view(self, kept_irreps) -> bool:
        """When the consumer of this layer reads only SOME of its output irreps (the reference's last conv layer emits
        all of conv_layer_irreps, model_factory/tfn_scalar_tensor.py:122-131, but the head that follows is an o3.Linear
        onto 16x0e+2x2e+4e: every other output irrep, the tensor-product paths that end in it and their radial-weight
        columns are computed and never read), an inference forward can run the layer for the kept irreps only.  The
        view is a second PointConv planned for ``kept_irreps`` that owns NO parameters: lin1 and the hidden radial
        layers are this layer's modules, the self-connection, lin2 and last radial layer read index-selected copies of
        this layer's parameters (same instruction blocks, same fan-in normalisation: every path into a kept output is
        kept).  Returns False (no view) when the irreps cannot be matched one to one."""
        kept = Irreps(kept_irreps).simplify()
        full_out = self.sc.irreps_out
        if any(sum(1 for _, ir2 in irr if ir2 == ir) != 1 for irr in (full_out, self.lin2.irreps_in) for _, ir in irr):
            return False
        if any(sum(1 for m2, ir2 in full_out if ir2 == ir and m2 == m) != 1 for m, ir in kept) or kept.dim >= full_out.dim:
            return False
        try:
            v = PointConv(conv_layer_irreps=kept, **self._ctor)
        except Exception:  # noqa: BLE001  (e.g. no tensor-product path into the kept irreps)
            return False

        from .. import plan as _plan

        m_sc, m_l2 = _plan.linear_flat_submap(v.sc.plan, self.sc.plan), _plan.linear_flat_submap(v.lin2.plan, self.lin2.plan)
        fullp = {(q.i_in1, q.i_sh, q.l3, q.p3): q for q in self.tp.plan.paths}
        cols = []
        for q in v.tp.plan.paths:
            f = fullp.get((q.i_in1, q.i_sh, q.l3, q.p3))
            if f is None or f.mul != q.mul:
                return False
            cols.append(np.arange(f.w_off, f.w_off + f.mul))
        if m_sc is None or m_l2 is None or not cols:
            return False
        v.lin1 = self.lin1
        mlp = v.tp.weight_nn
        mlp.layer0, mlp.layer1 = self.tp.weight_nn.layer0, self.tp.weight_nn.layer1
        mlp.__dict__["_hidden_from"] = self.tp.weight_nn
        for mod in (v.sc, v.lin2, mlp.layer2):
            del mod._parameters["weight"]
        v.__dict__["_slices"] = (WeightSlice(m_sc), WeightSlice(m_l2), WeightSlice(np.concatenate(cols), dim=1))
        self.__dict__["_view"] = v
        return True

    def _view_forward(self, data, differentiable: bool):
        v = self._view
        s_sc, s_l2, s_w2 = v._slices
        # inference: cached copies that follow the parameters' versions; under autograd: index_select nodes, so the kept
        # blocks' gradients reach the layer's parameters and everything else gets exact zeros -- what the reference's
        # autograd gives the weights of paths that never reach the loss
        pick = (lambda sl, w: sl.select(w)) if differentiable else (lambda sl, w: sl.get(w))
        # (plain instance attributes: Module.__setattr__ would register a Parameter that select() hands back unchanged)
        targets = ((v.sc, s_sc, self.sc.weight), (v.lin2, s_l2, self.lin2.weight),
                   (v.tp.weight_nn.layer2, s_w2, self.tp.weight_nn.layer2.weight))
        for mod, sl, w in targets:
            mod.__dict__["weight"] = pick(sl, w)
        try:
            return v(data)
        finally:
            if differentiable:   # do not keep autograd graphs alive through module attributes
                for mod, _, _ in targets:
                    mod.__dict__["weight"] = None

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        if self._view is not None and DEAD_PATH_ELIMINATION:
            # the consumer (model_factory.eliminate_dead_outputs paired it with this layer) takes the kept irreps only
            data = self._view_forward(data, _ag.needs_grad_lazy(lambda: (data[DataKey.NODE_FEATURES], *_ag.params_of(self))))
            data[KEPT_ONLY] = True
            return data
        data.pop(KEPT_ONLY, None)   # (a re-used batch dict may carry the marker of an earlier forward)
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
        # (small batches keep the mul_ir row + row kernels: matten_agg_linear's per-workgroup set-up -- weight fragments
        # and tables into LDS, species lookup -- is ~35 us of latency per launch that only a long stream pays back:
        # n100, 473 rows: 1.25 vs 1.08 ms per forward; 64 000 rows: -3 %)
        if (self.agg_plan is not None and x1.shape[0] >= AGG_KM_MIN_ROWS
                and not _ag.needs_grad_lazy(lambda: (x1, self_connection, self.lin2.weight, *_ag.params_of(self.tp.weight_nn)))):
            ap, t, dev = self.agg_plan, self._agg_tables, x1.device
            agg = self.tp(x1, data, self.avg_num_neighbors, out_layout=(t.get("entries", dev), ap.ld))
            gate = self.__dict__.get("_gate_fuse")   # set per call by PointConvWithActivation: (cmeta, act_cst, d_act, bn)
            if gate is not None:
                cmeta, act_cst, d_act, bn_scale, bn_shift = gate
                data[DataKey.NODE_FEATURES] = ops.agg_linear_gate(
                    agg, species, self._agg_wtab.get(self.lin2.weight), t.get("io", dev), t.get("blocks", dev), ap.d_out,
                    cmeta, act_cst, d_act, add=self_connection, bn_scale=bn_scale, bn_shift=bn_shift)
                data[GATE_APPLIED] = True
                return data
            data[DataKey.NODE_FEATURES] = ops.agg_linear(agg, species, self._agg_wtab.get(self.lin2.weight),
                                                         t.get("io", dev), t.get("blocks", dev), ap.d_out,
                                                         add=self_connection)
            return data
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

    def _gate_fuse_args(self, dev):
        """(cmeta, act_cst, d_act, bn_scale, bn_shift) for matten_agg_linear_gate, or None when this layer / mode keeps
        the separate Gate kernel: training-mode BatchNorm, instance normalisation, the norm activation, or a layer
        whose gates do not fit the kernel's register sets (plan.plan_agg_gate)."""
        if not GATE_FUSE or self.conv.agg_plan is None or getattr(self.act, "activation_type", "gate") != "gate":
            return None
        if self.norm.method not in ("batch", "none", None):
            return None
        bn = self.norm.n
        if bn is not None and bn.training:
            return None
        fuse = self.__dict__.get("_fuse_tables")
        if fuse is None:
            from .. import plan as _plan

            cm = _plan.plan_agg_gate(self.conv.agg_plan, self.act.plan)
            fuse = False if cm is None else DeviceTables(cmeta=cm)
            self.__dict__["_fuse_tables"] = fuse
            self.__dict__["_fuse_bn"] = DerivedWeight(self._fold_bn)
        if fuse is False:
            return None
        scale = shift = None
        if bn is not None:
            scale, shift = self._fuse_bn.get(bn.running_mean, bn.running_var, bn.weight, bn.bias)
        return (fuse.get("cmeta", dev), self.act._tables.get("act_cst", dev), self.act.plan.irreps_out.dim, scale, shift)

    def _fold_bn(self, rm, rv, w, b):
        """eval-mode BatchNorm as per-column (scale, shift) of the activated row (same folding as matten_gate_bn)"""
        meta = torch.as_tensor(np.asarray(self.act.plan.meta).reshape(-1, 4)[:, 3].astype(np.int64), device=w.device)
        bn_idx, mean_idx = meta & 0xFFFF, (meta >> 16) & 0xFFFF
        scale = (w / torch.sqrt(rv + self.norm.n.eps))[bn_idx]
        has_mean = mean_idx != 0xFFFF
        mi = torch.where(has_mean, mean_idx, torch.zeros_like(mean_idx))
        shift = torch.where(has_mean, b[mi] - rm[mi] * scale, torch.zeros_like(scale))
        return scale.contiguous(), shift.contiguous()

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        x = data[DataKey.NODE_FEATURES]
        fuse = None
        if not _ag.needs_grad_lazy(lambda: (x, *_ag.params_of(self.conv))) and x.shape[0] >= AGG_KM_MIN_ROWS:
            fuse = self._gate_fuse_args(x.device)
        self.conv.__dict__["_gate_fuse"] = fuse
        try:
            data = self.conv(data)
        finally:
            self.conv.__dict__["_gate_fuse"] = None
        if data.pop(GATE_APPLIED, False):
            return data   # lin2's kernel wrote the activated (and normalised) row
        # Gate and (eval-mode) BatchNorm run as one elementwise kernel
        data[DataKey.NODE_FEATURES] = self.act(data[DataKey.NODE_FEATURES], self.norm, data)
        return data
