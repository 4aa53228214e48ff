"""
torch.autograd bindings of the HIP operators for the training step.

The reference trains by autograd through e3nn's einsums and torch_scatter (model/model.py:276-372).  Here
each operator's adjoint is a hand-written kernel (matten_amd/csrc/backward.hip); autograd only chains them
and carries the cheap differentiable re-packings of the parameters (index + scale).  The radial MLP runs forward
(matten_radial_mlp) and backward (matten_radial_mlp_bwd) on this library's own fp32-MFMA kernels; no library GEMM is
left on the step (the 16-wide species embedding is an index lookup).
"""
from typing import Optional

import os

import torch

from . import ops


class SpeciesEmbedFn(torch.autograd.Function):
    """node features of the one-hot embedding, ``feats`` as the species_embed kernel computed them; the adjoint sums the
    incoming gradient per species with the weight-gradient kernel of the species linear (input = the constant 1:
    dWp[s, w] = sum over the species' rows of dy[n, w]; ordered partial sums, no atomics, no index sort)."""

    @staticmethod
    def forward(ctx, weight, bias, feats, species_order, segs):
        ctx.species_order, ctx.segs = species_order, segs   # segs: int32 [[0, 1, 1, 0, dim, 0, 0, 0]] on the device
        ctx.n_species = weight.shape[1]
        return feats.view_as(feats)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, dim = g.shape
        ones = torch.ones(n, 1, dtype=torch.float32, device=g.device)
        per_species = ops.species_linear_wgrad(ones, g, ctx.species_order, ctx.n_species, [ctx.segs], dim)   # [S, dim]
        return per_species.t(), per_species.sum(0), None, None, None


class SpeciesLinearFn(torch.autograd.Function):
    """out = add + x W_species  (matten_species_linear) from the flat e3nn ``weight``.
    The per-species packing packed[s, j] = weight[gather[s, j]] * scale[j] is part of this node (one autograd node and
    one Function.apply per linear instead of two: the eager step is bound by that host work at small batches): forward
    packs (and, when x needs a gradient, emits the per-path transposed packing from the same launch), backward runs
    the same kernel with W^T for dx, the weight-gradient kernel for the packed gradient, and maps that back through
    the inverse permutation (``gather`` is a bijection: no index sort, no atomics)."""

    @staticmethod
    def forward(ctx, x, weight, add, mod, species_order):
        plan, dev, t = mod.plan, x.device, mod._tables
        gather, scale = t.get("gather", dev), t.get("scale", dev)
        if ctx.needs_input_grad[0]:
            wp, wpt = ops.gather_scale(weight, gather, scale, perm2=t.get("perm_t", dev))
        else:
            wp, wpt = ops.gather_scale(weight, gather, scale), None
        segs = [t.get(f"meta{i}", dev) for i in range(len(plan.passes))]
        ctx.mod, ctx.species_order, ctx.wpt = mod, species_order, wpt
        ctx.save_for_backward(x, wp)
        ctx.has_add = add is not None
        return ops.species_linear(x, species_order, wp, plan.w_stride, segs, plan.d_out, add, plan.fully_covered)

    @staticmethod
    def backward(ctx, g):
        x, wp = ctx.saved_tensors
        mod, plan, dev = ctx.mod, ctx.mod.plan, g.device
        t = mod._tables
        g = g.contiguous()
        dx = dweight = None
        if ctx.needs_input_grad[0]:
            segs_t = [t.get(f"meta_t{i}", dev) for i in range(len(plan.passes_t))]
            dx = ops.species_linear(g, ctx.species_order, ctx.wpt, plan.w_stride, segs_t, plan.d_in, None,
                                    plan.input_covered)
        if ctx.needs_input_grad[1]:
            segs = [t.get(f"meta{i}", dev) for i in range(len(plan.passes))]
            n_species = wp.shape[0] if wp.dim() == 2 else 1
            dwp = ops.species_linear_wgrad(x, g, ctx.species_order, n_species, segs, plan.w_stride)
            dweight = ops.gather_scale(dwp.reshape(-1), t.get("gather_inv", dev), t.get("scale", dev), scale_by_source=True)
        return dx, dweight, (g if ctx.has_add else None), None, None


# Storage type of the two per-edge tensors of a training step, the radial weights w[E, W] and their gradient -- at
# large batches the step's dominant memory traffic (BASELINE.json configs[3] names bf16 storage; the reference itself
# is fp32 throughout, data/_dtype.py:3).  bf16 is opt-in (MATTEN_EDGE_STORAGE=bf16 or set_edge_storage_dtype): the MLP
# kernel rounds w to bf16 on the store (nearest even), every kernel computes and accumulates in fp32, node features,
# neighbour sums, BatchNorm statistics and all parameters stay fp32.
EDGE_STORAGE_DTYPE = torch.bfloat16 if os.environ.get("MATTEN_EDGE_STORAGE", "fp32").lower() == "bf16" else torch.float32


def set_edge_storage_dtype(dtype) -> None:
    global EDGE_STORAGE_DTYPE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("edge storage is fp32 or bf16")
    EDGE_STORAGE_DTYPE = dtype


class RadialMLPFn(torch.autograd.Function):
    """w[E, w_pad] = MLP(bessel(|edge|)) in the reference's weight-column order (pad columns zero); the adjoint maps
    dL/dw back to the three raw weight matrices (the packing is a per-layer scale and a row padding)."""

    @staticmethod
    def forward(ctx, w0, w1, w2, mod, geom_sorted, n_basis, r_start, r_end):
        w0p, w1p, w2p = ops.radial_pack(w0, w1, w2, mod.pack_scales())          # one launch for the three layers
        ctx.mod, ctx.rbf = mod, (int(n_basis), float(r_start), float(r_end))
        ctx.save_for_backward(geom_sorted, w0p, w1p, w2p)
        return ops.radial_mlp(geom_sorted, int(n_basis), float(r_start), float(r_end), w0p, w1p, w2p,
                              out_dtype=EDGE_STORAGE_DTYPE)

    @staticmethod
    def backward(ctx, g):
        geom, w0p, w1p, w2p = ctx.saved_tensors
        mod = ctx.mod
        nb, W = mod.hs[0], mod.hs[3]
        # the kernels multiply their partial sums by the packing factors: gradients w.r.t. the raw layers come back
        d0, d1, d2 = ops.radial_mlp_bwd(geom, *ctx.rbf, w0p, w1p, w2p, W, g.contiguous(), scales=mod.pack_scales())
        return d0[:nb], d1, d2[:, :W], None, None, None, None, None


# training forward: batches below this many nodes walk their CSR segments in pieces (see TensorProductScatterFn.forward)
TRAIN_HUB_SPLIT_MAX_ROWS = int(os.environ.get("MATTEN_HUB_SPLIT_MAX_ROWS_TRAIN", "65536"))


class TensorProductScatterFn(torch.autograd.Function):
    """agg = sum_{edges -> n} uvu(x[src], Y, w) * norm   with w[E,W] in the reference column order."""

    @staticmethod
    def forward(ctx, x, w_edge, mod, data, avg, num_neigh):
        from .data.irreps import DataKey

        dev = x.device
        p = mod.plan
        ctx.mod, ctx.avg, ctx.num_neigh = mod, avg, num_neigh
        ctx.graph = (data[DataKey.AMD_SH], data[DataKey.AMD_SRC], data["_amd_dst_sorted"])
        ctx.out_csr = data.get("_amd_out_csr")
        ctx.save_for_backward(x, w_edge)
        # the forward walks a destination node's CSR segment serially: irregular batches (real crystals: 30 neighbours on
        # average, 80 in the one-atom cells) are cut into pieces of <= HUB_SPLIT_LEN edges and summed in order afterwards,
        # like the inference forward of small batches (nn/utils.py); the adjoint is per edge and does not care
        from .nn import utils as nnu

        rowptr, split = data[DataKey.AMD_ROWPTR], None
        if nnu.HUB_SPLIT_LEN > 0 and x.shape[0] < TRAIN_HUB_SPLIT_MAX_ROWS:
            # (its own cache key: the inference forward cuts the same rowptr into pieces of another length)
            key = ("_amd_csr_split", nnu.HUB_SPLIT_LEN, num_neigh is None)
            split = data.get(key)
            if split is None or split[3] is not rowptr:
                split = ops.csr_split(rowptr, data[DataKey.AMD_SRC].shape[0], nnu.HUB_SPLIT_LEN, num_neigh) + (rowptr,)
                data[key] = split
        agg = ops.tp_paths(x, w_edge, data[DataKey.AMD_SH], rowptr if split is None else split[0], data[DataKey.AMD_SRC],
                           mod._tables.get("entries", dev), mod._tables.get("unit_start", dev), p.units_per_tile,
                           p.d_mid, avg, num_neigh if split is None else split[2])
        return agg if split is None else ops.segment_reduce(agg, split[1], mean=False)

    @staticmethod
    def backward(ctx, g):
        x, w_edge = ctx.saved_tensors
        mod, dev = ctx.mod, g.device
        sh, src, dst = ctx.graph
        impl = os.environ.get("MATTEN_TP_BWD", "lit")   # "lit": literal-coefficient kernel; "table": the table-driven ones
        if impl == "lit":
            # dx summed per source node in a fixed order (MATTEN_TP_BWD_DX=atomic: fp32 atomics, order not fixed)
            out_csr = ctx.out_csr if os.environ.get("MATTEN_TP_BWD_DX", "ordered") != "atomic" else None
            dx, dw = ops.tp_backward_lit(x, w_edge, sh, src, dst, mod._tables.get("bw_blocks", dev),
                                         mod._tables.get("bw_paths", dev), mod.plan.bw_sum_lanes, g.contiguous(), ctx.avg,
                                         ctx.num_neigh, out_csr=out_csr, max_l=mod.plan.bw_max_l,
                                         blocks_cover_input=int(mod.plan.bw_blocks[:, 1].dot(2 * mod.plan.bw_blocks[:, 2] + 1))
                                         == mod.plan.d_in)
            return dx, dw, None, None, None, None
        dx, dw = ops.tp_backward(x, w_edge, sh, src, dst, mod._tables.get("bw_col_meta", dev),
                                 mod._tables.get("bw_nnz_ijk", dev), mod._tables.get("bw_nnz_c", dev), g.contiguous(),
                                 ctx.avg, ctx.num_neigh,
                                 in_groups=None if os.environ.get("MATTEN_TP_BWD_GROUPED", "1") == "0" else
                                 (mod._tables.get("bw_in_ptr", dev), mod._tables.get("bw_in_cols", dev)))
        return dx, dw, None, None, None, None


# training on the fused forward: the adjoint re-evaluates w inside its workgroups (0 = matten_radial_mlp writes it for one layer
# at a time inside the backward, the round-3 form)
W_FREE_ADJOINT = os.environ.get("MATTEN_TP_BWD_WFREE", "1") != "0"


class FusedTensorProductFn(torch.autograd.Function):
    """The training tensor product on the PRODUCTION kernel (MATTEN_TRAIN_TP=fused): forward = matten_tp_fused -- the last
    radial layer on the matrix cores inside the kernel, the per-edge weights w[E, W] never reach memory -- and nothing
    per-edge is kept for the backward except what the batch already holds (geometry, harmonics).  The backward
    re-evaluates w for THIS layer with matten_radial_mlp (0.1 ms per layer at 290 k edges), runs the literal adjoint and
    the radial MLP's adjoint, and drops w again: w exists for one layer at a time, inside the backward only (at batch
    2048 that is 4 x 1 GB less live memory between the passes).  The fused kernel's operands (weights in fused column
    order, range scale, fp16 hi/lo fragments) come from the raw layers through three small kernels
    (ops.fused_operands), so the step stays capturable."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2, mod, data, avg, num_neigh):
        from .data.irreps import DataKey

        dev = x.device
        p, mlp = mod.plan, mod.weight_nn
        nb, r0, r1 = data[DataKey.AMD_RBF].tolist()
        ctx.mod, ctx.avg, ctx.num_neigh = mod, avg, num_neigh
        ctx.rbf = (int(nb), float(r0), float(r1))
        ctx.graph = (data[DataKey.AMD_GEOM], data[DataKey.AMD_SH], data[DataKey.AMD_SRC], data["_amd_dst_sorted"])
        ctx.out_csr = data.get("_amd_out_csr")
        ctx.save_for_backward(x, w0, w1, w2)
        # the fused kernel's operands from the raw layers: four small launches, nothing read on the host (the inference
        # path derives them once with library ops and caches them; here the parameters change every step)
        gent = mod._tables.get("gentries", dev)
        w0p, w1p, w2p, hs, frag, inv = ops.fused_operands(w0, w1, w2, mlp.pack_scales(), mod._tables.get("fused_cols", dev),
                                                          gent, p.fused_a_tiles, r0, r1, mlp.act_cst)
        h2p = ops.radial_hidden(data[DataKey.AMD_GEOM], int(nb), r0, r1, w0p, w1p, hs)
        # the hidden features (128 B per edge) are what the backward re-evaluates w from; a layer with an input block wider
        # than the w-free kernel's workgroup (ops.WFREE_MAX_MUL channels) keeps the adjoint on a materialised w instead
        if W_FREE_ADJOINT and p.bw_max_mul <= ops.WFREE_MAX_MUL:
            ctx.h2p, ctx.hs = h2p, hs
        return ops.tp_fused(x, h2p, w2p, data[DataKey.AMD_SH], data[DataKey.AMD_ROWPTR], data[DataKey.AMD_SRC], gent,
                            mod._tables.get("gumap", dev), len(p.fused_unit_map), p.fused_lds_floats_per_wave, p.d_mid, avg,
                            num_neigh, a_split=(frag, inv))

    @staticmethod
    def backward(ctx, g):
        x, w0, w1, w2 = ctx.saved_tensors
        mod, dev = ctx.mod, g.device
        mlp = mod.weight_nn
        geom, sh, src, dst = ctx.graph
        scales = mlp.pack_scales()
        w0p, w1p, w2p = ops.radial_pack(w0, w1, w2, scales)                       # reference column order
        out_csr = ctx.out_csr if os.environ.get("MATTEN_TP_BWD_DX", "ordered") != "atomic" else None
        covered = int(mod.plan.bw_blocks[:, 1].dot(2 * mod.plan.bw_blocks[:, 2] + 1)) == mod.plan.d_in
        h2p = getattr(ctx, "h2p", None)
        if h2p is not None:
            # w-free: every workgroup of the adjoint re-evaluates its weights on the matrix cores from the forward's hidden
            # features and the last layer's A fragments in reference column order (one small launch); w never exists
            frag, inv = ops.split_a_tiles_dev(w2p, mod._tables.get("bw_w_entries", dev), mod.plan.bw_a_tiles, ctx.hs)
            dx, dw = ops.tp_backward_lit(x, None, sh, src, dst, mod._tables.get("bw_blocks", dev),
                                         mod._tables.get("bw_paths", dev), mod.plan.bw_sum_lanes, g.contiguous(), ctx.avg,
                                         ctx.num_neigh, out_csr=out_csr, blocks_cover_input=covered, wfree=(h2p, frag, inv),
                                         dw_shape=((geom.shape[0], w2p.shape[1]), EDGE_STORAGE_DTYPE),
                                         lds_floats=mod.plan.bw_wfree_lds_floats, max_l=mod.plan.bw_max_l,
                                         max_mul=mod.plan.bw_max_mul)
            # (h2p stays on the ctx until autograd releases it: a second backward under retain_graph takes the same route)
        else:
            w_edge = ops.radial_mlp(geom, *ctx.rbf, w0p, w1p, w2p, out_dtype=EDGE_STORAGE_DTYPE)   # transient
            dx, dw = ops.tp_backward_lit(x, w_edge, sh, src, dst, mod._tables.get("bw_blocks", dev),
                                         mod._tables.get("bw_paths", dev), mod.plan.bw_sum_lanes, g.contiguous(), ctx.avg,
                                         ctx.num_neigh, out_csr=out_csr, blocks_cover_input=covered, max_l=mod.plan.bw_max_l)
            del w_edge
        nb, W = mlp.hs[0], mlp.hs[3]
        d0, d1, d2 = ops.radial_mlp_bwd(geom, *ctx.rbf, w0p, w1p, w2p, W, dw, scales=scales)
        return dx, d0[:nb], d1, d2[:, :W], None, None, None, None


class GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        dev = x.device
        ctx.mod = mod
        ctx.save_for_backward(x)
        return ops.gate_bn(x, mod._tables.get("meta", dev), mod._tables.get("act_cst", dev))

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mod, dev = ctx.mod, g.device
        return ops.gate_bwd(x, mod._tables.get("meta", dev), mod._tables.get("act_cst", dev), g.contiguous()), None


class NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        ctx.mod = mod
        ctx.save_for_backward(x)
        return ops.norm_act(x, mod._tables.get("chan", x.device), mod.plan.act_code, mod.plan.epsilon)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        return ops.norm_act_bwd(x, g.contiguous(), mod._tables.get("chan", g.device), mod.plan.act_code,
                                mod.plan.epsilon), None


class BatchNormTrainFn(torch.autograd.Function):
    """e3nn BatchNorm with batch statistics; the running averages are updated in place by the statistics kernel
    (running = (1 - momentum) running + momentum batch), so the step needs no small library launches for them."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        dev = x.device
        y, mean, nu = ops.bn_train_fwd(x, bn._tables.get("col2chan", dev), bn._tables.get("chan", dev), weight, bias,
                                       bn.eps, bn.running_mean, bn.running_var, bn.momentum)
        ctx.bn = bn
        ctx.save_for_backward(x, weight, mean, nu)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, nu = ctx.saved_tensors
        bn, dev = ctx.bn, g.device
        dx, dweight, dbias = ops.bn_train_bwd(x, g.contiguous(), bn._tables.get("col2chan", dev),
                                              bn._tables.get("chan", dev), mean, nu, weight, bn.eps, bn.bias.numel())
        return dx, dweight, dbias, None


class InstanceNormFn(torch.autograd.Function):
    """the reference's InstanceNorm (per-crystal statistics; same arithmetic as BatchNormTrainFn per crystal)"""

    @staticmethod
    def forward(ctx, x, weight, bias, norm, ptr, batch):
        dev = x.device
        y, mean, nu = ops.instance_norm_fwd(x, ptr, batch, norm._tables.get("col2chan", dev), norm._tables.get("chan", dev),
                                            weight, bias, norm.eps)
        ctx.norm = norm
        ctx.save_for_backward(x, weight, mean, nu, ptr, batch)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, nu, ptr, batch = ctx.saved_tensors
        norm, dev = ctx.norm, g.device
        dx, A, B = ops.instance_norm_bwd(x, g.contiguous(), ptr, batch, norm._tables.get("col2chan", dev),
                                         norm._tables.get("chan", dev), mean, nu, weight, norm.eps)
        dweight = (A * torch.rsqrt(nu + norm.eps)).sum(0)
        dbias = B.sum(0)[norm._tables.get("scalar_chan", dev)]
        return dx, dweight, dbias, None, None, None


class SegmentReduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr, mean):
        ctx.ptr, ctx.mean, ctx.n = ptr, mean, x.shape[0]
        return ops.segment_reduce(x, ptr, mean)

    @staticmethod
    def backward(ctx, g):
        return ops.segment_reduce_bwd(g.contiguous(), ctx.ptr, ctx.n, ctx.mean), None, None


class SegmentMinMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr, take_max):
        out, arg = ops.segment_minmax(x, ptr, take_max, want_arg=True)
        ctx.n = x.shape[0]
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return ops.segment_minmax_bwd(g.contiguous(), arg, ctx.n), None, None


def needs_grad(*tensors: Optional[torch.Tensor]) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def params_of(module) -> tuple:
    """the module's parameters as a tuple, built once (module.parameters() walks the module tree: ~15 us per call for a
    conv layer, called several times per forward).  The Parameter objects are stable -- FlatAdam re-homes their storage,
    not the objects -- and their requires_grad flags are read live."""
    cached = module.__dict__.get("_amd_params")
    if cached is None:
        cached = tuple(module.parameters())
        module.__dict__["_amd_params"] = cached
    return cached


def needs_grad_lazy(tensors) -> bool:
    """needs_grad over the tuple ``tensors()`` returns, which is only built with autograd on (walking a module's
    parameters() a dozen times per forward is 0.1 ms of host time that an inference forward of a small batch feels)"""
    return torch.is_grad_enabled() and needs_grad(*tensors())
