"""
Thin tensor-level wrappers over the C ABI (include/matten_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; all arithmetic happens in
libmatten_hip.so.  Every wrapper insists on contiguous CUDA(=HIP) tensors of the right dtype and
raises if handed anything else -- there is deliberately no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import numpy as np

import torch

from . import _lib
from .nn._tables import bump_weights_epoch


# ---- optional HIP-event timing of individual launches (used by bench.py for the roofline line) ----
_EVENTS = None  # name -> list of (start_event, end_event) recorded on the launch stream
_EVENTS_ONLY = None  # name prefixes to time, or None: every timed launch


def enable_event_timing(flag: bool = True, only=None) -> None:
    """only: time just the launches whose name starts with one of these prefixes (an event pair costs a few us of queue
    time per launch: a benchmark that needs one kernel's duration should not pay for nine)"""
    global _EVENTS, _EVENTS_ONLY
    _EVENTS = {} if flag else None
    _EVENTS_ONLY = tuple(only) if (flag and only) else None


def event_timings_ms():
    """name -> list of elapsed milliseconds (call after torch.cuda.synchronize())."""
    if _EVENTS is None:
        return {}
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _EVENTS.items()}


class _timed:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.on = _EVENTS is not None and (_EVENTS_ONLY is None or self.name.startswith(_EVENTS_ONLY))
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.on and _EVENTS is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream())
            _EVENTS.setdefault(self.name, []).append((self.a, b))
        return False


SH_STRIDE = 32  # floats per row of sh_sorted


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """the current HIP stream of the current device as a raw handle (the private getter skips the Stream object that
    torch.cuda.current_stream() builds: ~4 us per launch, 40+ launches per forward)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.MattenHipError(
            f"{name} is on {t.device}: matten_amd runs on MI355X only (no CPU fallback). Move the batch to cuda."
        )
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _need_rows(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    """like _need, but a column slice of a wider row-major matrix (unit column stride) is passed through as is: the
    kernels that take it address rows by the tensor's own row stride"""
    if isinstance(t, torch.Tensor) and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.is_cuda \
            and t.dtype == dtype:
        return t
    return _need(t, dtype, name)


def _err_flag(err, dev) -> torch.Tensor:
    """a zeroed int32[1] flag word, or the caller's (one allocation + one memset for several kernels' flags)"""
    if err is None:
        return torch.zeros(1, dtype=torch.int32, device=dev)
    if err.dtype != torch.int32 or err.numel() != 1 or err.device != dev:
        raise ValueError("err must be an int32 tensor of one element on the inputs' device")
    return err


def csr_build(edge_index: torch.Tensor, n_nodes: int, err: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (perm[E] i32, rowptr[N+1] i32, src_sorted[E] i32, err_flag[1] i32)"""
    lib = _lib.load()
    edge_index = _need(edge_index, torch.int64, "edge_index")
    E = edge_index.shape[1]
    dev = edge_index.device
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    src = torch.empty(E, dtype=torch.int32, device=dev)
    err = _err_flag(err, dev)
    nbytes = lib.matten_csr_workspace_bytes(E, n_nodes)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    _lib.check(
        lib.matten_csr_build(_ptr(edge_index), E, n_nodes, _ptr(perm), _ptr(rowptr), _ptr(src), _ptr(ws), nbytes,
                             _ptr(err), _stream()),
        "matten_csr_build",
    )
    return perm, rowptr, src, err


def csr_split(rowptr: torch.Tensor, n_edges: int, max_len: int, num_neigh: torch.Tensor = None):
    """-> (vrowptr[bound+1] i32, vseg[N+1] i64, vnn[bound] f32 or None): every CSR segment cut into virtual nodes of at
    most max_len edges (include/matten_hip.h matten_csr_split); bound = N + E // max_len, no host sync"""
    lib = _lib.load()
    rowptr = _need(rowptr, torch.int32, "rowptr")
    N = rowptr.shape[0] - 1
    bound = int(lib.matten_csr_split_bound(N, n_edges, max_len))
    dev = rowptr.device
    vrowptr = torch.empty(bound + 1, dtype=torch.int32, device=dev)
    vseg = torch.empty(N + 1, dtype=torch.int64, device=dev)
    vnn = None
    if num_neigh is not None:
        num_neigh = _need(num_neigh, torch.float32, "num_neigh")
        vnn = torch.empty(bound, dtype=torch.float32, device=dev)
    _lib.check(lib.matten_csr_split(_ptr(rowptr), N, n_edges, max_len, _ptr(vrowptr), _ptr(vseg), _ptr(num_neigh), _ptr(vnn),
                                    _stream()), "matten_csr_split")
    return vrowptr, vseg, vnn


def group_by_key(key: torch.Tensor, n_keys: int, err: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (order[n] i32: positions stably sorted by key, seg[n_keys+1] i32, err_flag[1] i32)"""
    lib = _lib.load()
    key = _need(key, torch.int64, "key")
    n = key.shape[0]
    dev = key.device
    order = torch.empty(n, dtype=torch.int32, device=dev)
    seg = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    err = _err_flag(err, dev)
    nbytes = lib.matten_group_workspace_bytes(n, n_keys)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.matten_group_by_key(_ptr(key), n, n_keys, _ptr(order), _ptr(seg), _ptr(ws), nbytes, _ptr(err),
                                       _stream()), "matten_group_by_key")
    return order, seg, err


def species_embed(atomic_numbers, z_to_index, min_z: int, max_z: int, n_species: int, weight, bias,
                  want_attrs: bool = False, err: torch.Tensor = None):
    """-> (species_index i64 [N], species_i32 [N], node_feats [N,dim], node_attrs [N,S] | None, err_flag)"""
    lib = _lib.load()
    Z = _need(atomic_numbers, torch.int64, "atomic_numbers")
    lut = _need(z_to_index, torch.int64, "_Z_to_index")
    W = _need(weight, torch.float32, "linear.weight")
    b = _need(bias, torch.float32, "linear.bias")
    N, dim = Z.shape[0], W.shape[0]
    dev = Z.device
    sidx = torch.empty(N, dtype=torch.int64, device=dev)
    s32 = torch.empty(N, dtype=torch.int32, device=dev)
    feats = torch.empty(N, dim, dtype=torch.float32, device=dev)
    attrs = torch.empty(N, n_species, dtype=torch.float32, device=dev) if want_attrs else None
    err = _err_flag(err, dev)
    _lib.check(
        lib.matten_species_embed(_ptr(Z), N, _ptr(lut), min_z, max_z, n_species, _ptr(W), _ptr(b), dim, _ptr(sidx),
                                 _ptr(s32), _ptr(feats), _ptr(attrs), _ptr(err), _stream()),
        "matten_species_embed",
    )
    return sidx, s32, feats, attrs, err


def edge_geom(pos, edge_index, edge_cell_shift, cell, batch, perm, lmax: int, n_basis: int = 0, r_start: float = 0.0,
              r_end: float = 1.0, want_vectors=False, want_lengths=False, want_attrs=False, want_embedding=False):
    """-> dict(geom_sorted [E,4], sh_sorted [E,(lmax+1)^2], + requested original-order tensors)"""
    lib = _lib.load()
    pos = _need(pos, torch.float32, "pos")
    edge_index = _need(edge_index, torch.int64, "edge_index")
    E = edge_index.shape[1]
    dev = pos.device
    n_cells = 0
    if cell is not None:
        cell = _need(cell, torch.float32, "cell").reshape(-1, 3, 3)
        n_cells = cell.shape[0]
        edge_cell_shift = _need(edge_cell_shift, torch.float32, "edge_cell_shift")
        if n_cells > 1:
            batch = _need(batch, torch.int64, "batch")
    sh_dim = (lmax + 1) ** 2
    geom = torch.empty(E, 4, dtype=torch.float32, device=dev)
    # rows padded to 32 floats: one 128-byte line per edge, and the fused TP kernel may read any l2 <= 4
    # (the kernel writes the padding columns as zeros itself)
    sh = torch.empty(E, SH_STRIDE, dtype=torch.float32, device=dev)
    out = {"geom_sorted": geom, "sh_sorted": sh}
    ev = torch.empty(E, 3, dtype=torch.float32, device=dev) if want_vectors else None
    el = torch.empty(E, dtype=torch.float32, device=dev) if want_lengths else None
    ea = torch.empty(E, sh_dim, dtype=torch.float32, device=dev) if want_attrs else None
    ee = torch.empty(E, n_basis, dtype=torch.float32, device=dev) if want_embedding else None
    _lib.check(
        lib.matten_edge_geom(_ptr(pos), _ptr(edge_index), _ptr(edge_cell_shift), _ptr(cell), n_cells, _ptr(batch),
                             _ptr(perm), E, pos.shape[0], lmax, n_basis, r_start, r_end, _ptr(geom), _ptr(sh), SH_STRIDE, _ptr(ev),
                             _ptr(el),
                             _ptr(ea), _ptr(ee), _stream()),
        "matten_edge_geom",
    )
    out.update(edge_vectors=ev, edge_lengths=el, edge_attrs=ea, edge_embedding=ee)
    return out


def _edge_dtype(t: torch.Tensor, name: str) -> torch.Tensor:
    """per-edge training tensors (radial weights, their gradient): fp32 or, opt-in, bf16 storage"""
    if not isinstance(t, torch.Tensor) or t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{name}: expected an fp32 or bf16 tensor")
    return _need(t, t.dtype, name)


def radial_mlp(geom_sorted, n_basis: int, r_start: float, r_end: float, w0p, w1p, w2p, out_dtype=torch.float32) -> torch.Tensor:
    lib = _lib.load()
    geom_sorted = _need(geom_sorted, torch.float32, "geom_sorted")
    w0p, w1p, w2p = (_need(w, torch.float32, n) for w, n in ((w0p, "w0p"), (w1p, "w1p"), (w2p, "w2p")))
    E = geom_sorted.shape[0]
    nb_pad, hidden = w0p.shape
    w_pad = w2p.shape[1]
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("radial_mlp: out_dtype must be fp32 or bf16")
    out = torch.empty(E, w_pad, dtype=out_dtype, device=geom_sorted.device)
    with _timed(f"radial_mlp/w_pad={w_pad}"):
        rc = lib.matten_radial_mlp(_ptr(geom_sorted), E, n_basis, r_start, r_end, _ptr(w0p), nb_pad, _ptr(w1p),
                                   _ptr(w2p), hidden, w_pad, 1.0, _ptr(out), int(out_dtype == torch.bfloat16), _stream())
    _lib.check(rc, "matten_radial_mlp")
    return out


def radial_pack(w0, w1, w2, scales):
    """raw radial layers -> (w0p [nb_pad,32], w1p [32,32], w2p [32,w_pad]) = the operands of radial_mlp in the reference's
    column order, each multiplied by its scale, in one launch"""
    lib = _lib.load()
    w0, w1, w2 = (_need(w, torch.float32, n) for w, n in ((w0, "layer0.weight"), (w1, "layer1.weight"), (w2, "layer2.weight")))
    nb, h = w0.shape
    W = w2.shape[1]
    nb_pad, w_pad = (nb + 3) // 4 * 4, (W + 15) // 16 * 16
    dev = w0.device
    w0p = torch.empty(nb_pad, h, dtype=torch.float32, device=dev)
    w1p = torch.empty(h, h, dtype=torch.float32, device=dev)
    w2p = torch.empty(h, w_pad, dtype=torch.float32, device=dev)
    _lib.check(lib.matten_radial_pack(_ptr(w0), _ptr(w1), _ptr(w2), nb, nb_pad, W, w_pad, float(scales[0]), float(scales[1]),
                                      float(scales[2]), _ptr(w0p), _ptr(w1p), _ptr(w2p), _stream()), "matten_radial_pack")
    return w0p, w1p, w2p


def fused_operands(w0, w1, w2, scales, cols, group_entries, n_tiles: int, r_start: float, r_end: float, act_cst: float):
    """Everything matten_radial_hidden + matten_tp_fused need, derived from the RAW radial layers by four small launches
    and no host reads (capturable): -> (w0p, w1p, w2p in fused column order, h_scale [2], a_split fragments, a_scale_inv)"""
    lib = _lib.load()
    w0, w1, w2 = (_need(w, torch.float32, n) for w, n in ((w0, "layer0.weight"), (w1, "layer1.weight"), (w2, "layer2.weight")))
    cols = _need(cols, torch.int64, "fused_cols")
    group_entries = _need(group_entries, torch.int32, "group_entries")
    nb, h = w0.shape
    W, n_cols = w2.shape[1], cols.numel()
    nb_pad, w_pad = (nb + 3) // 4 * 4, (n_cols + 15) // 16 * 16 + 16   # +16: whole tiles may start at any entry
    dev = w0.device
    w0p = torch.empty(nb_pad, h, dtype=torch.float32, device=dev)
    w1p = torch.empty(h, h, dtype=torch.float32, device=dev)
    w2p = torch.empty(h, w_pad, dtype=torch.float32, device=dev)
    hs = torch.empty(2, dtype=torch.float32, device=dev)
    n_ent = group_entries.shape[0]
    frag = torch.empty(n_tiles, 64, 16, dtype=torch.float16, device=dev)
    inv = torch.empty(n_ent, dtype=torch.float32, device=dev)
    st = _stream()
    _lib.check(lib.matten_radial_pack_cols(_ptr(w0), _ptr(w1), _ptr(w2), nb, nb_pad, W, _ptr(cols), n_cols, w_pad,
                                           float(scales[0]), float(scales[1]), float(scales[2]), _ptr(w0p), _ptr(w1p),
                                           _ptr(w2p), st), "matten_radial_pack_cols")
    _lib.check(lib.matten_radial_h_scale(_ptr(w0), _ptr(w1), nb, float(r_start), float(r_end), float(act_cst), _ptr(hs), st),
               "matten_radial_h_scale")
    _lib.check(lib.matten_split_a_tiles(_ptr(w2p), w_pad, _ptr(group_entries), n_ent, _ptr(hs), _ptr(frag), _ptr(inv), st),
               "matten_split_a_tiles")
    return w0p, w1p, w2p, hs, frag, inv


def gather_scale(src, idx, scale, scale_by_source: bool = False, perm2=None):
    """out[i] = src.flat[idx.flat[i]] * scale[(idx.flat[i] if scale_by_source else i) % len(scale)], shaped like idx.
    perm2 (int64 [len(scale)]): also return out2[r, q] = out[r, perm2[q]] (the same launch)"""
    lib = _lib.load()
    src = _need(src, torch.float32, "src")
    idx = _need(idx, torch.int64, "idx")
    scale = _need(scale, torch.float32, "scale")
    out = torch.empty(idx.shape, dtype=torch.float32, device=src.device)
    out2 = None
    if perm2 is not None:
        perm2 = _need(perm2, torch.int64, "perm2")
        out2 = torch.empty(idx.shape, dtype=torch.float32, device=src.device)
    _lib.check(lib.matten_gather_scale(_ptr(src), _ptr(idx), _ptr(scale), idx.numel(), scale.numel(), int(scale_by_source),
                                       _ptr(out), _ptr(perm2), _ptr(out2), _stream()), "matten_gather_scale")
    return out if perm2 is None else (out, out2)


def radial_mlp_bwd(geom_sorted, n_basis: int, r_start: float, r_end: float, w0p, w1p, w2p, w_cols: int, dw,
                   scales=(1.0, 1.0, 1.0)):
    """Adjoint of radial_mlp -> (dW0 [nb_pad,32], dW1 [32,32], dW2 [32,w_pad]): gradients w.r.t. the packed weights
    times `scales`, i.e. w.r.t. the raw layers when `scales` are the factors radial_pack applied.
    dw [E, ld >= w_pad] fp32 or bf16: dL/dw from tp_backward (pad columns are never read as data)."""
    lib = _lib.load()
    geom_sorted = _need(geom_sorted, torch.float32, "geom_sorted")
    w0p, w1p, w2p = (_need(w, torch.float32, n) for w, n in ((w0p, "w0p"), (w1p, "w1p"), (w2p, "w2p")))
    if dw.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("dw must be fp32 or bf16")
    dw = _need(dw, dw.dtype, "dw")
    E = geom_sorted.shape[0]
    nb_pad, hidden = w0p.shape
    w_pad = w2p.shape[1]
    dev = geom_sorted.device
    n_small, n_rng = lib.matten_radial_mlp_bwd_small_slices(E), lib.matten_radial_mlp_bwd_w2_ranges(E, w_pad)
    h2 = torch.empty(E, hidden, dtype=torch.float32, device=dev)
    part_small = torch.empty(max(n_small, 1), nb_pad * hidden + hidden * hidden, dtype=torch.float32, device=dev)
    part_w2 = torch.empty(max(n_rng, 1), hidden, w_pad, dtype=torch.float32, device=dev)
    if E == 0:
        return w0p.new_zeros(nb_pad, hidden), w1p.new_zeros(hidden, hidden), w2p.new_zeros(hidden, w_pad)
    small = torch.empty(nb_pad * hidden + hidden * hidden, dtype=torch.float32, device=dev)
    d2 = torch.empty(hidden, w_pad, dtype=torch.float32, device=dev)
    with _timed(f"radial_mlp_bwd/w_pad={w_pad}"):
        # (the per-wave / per-range partial sums are added in a fixed order by the call's last launch: no atomics)
        rc = lib.matten_radial_mlp_bwd(_ptr(geom_sorted), E, n_basis, r_start, r_end, _ptr(w0p), nb_pad, _ptr(w1p),
                                       _ptr(w2p), hidden, w_pad, int(w_cols), _ptr(dw), dw.shape[1],
                                       int(dw.dtype == torch.bfloat16), _ptr(h2), _ptr(part_small), _ptr(part_w2),
                                       float(scales[0]), float(scales[1]), float(scales[2]), _ptr(small), _ptr(d2),
                                       _stream())
    _lib.check(rc, "matten_radial_mlp_bwd")
    return small[: nb_pad * hidden].reshape(nb_pad, hidden), small[nb_pad * hidden:].reshape(hidden, hidden), d2


def tp_paths(x, w_edge, sh_sorted, rowptr, src_sorted, entries, unit_start, units_per_tile: int, d_mid: int,
             avg_num_neighbors: float, num_neigh=None) -> torch.Tensor:
    lib = _lib.load()
    from .plan import TP_TILE_NODES

    if lib.matten_tp_tile_nodes() != TP_TILE_NODES:
        raise _lib.MattenHipError("plan.TP_TILE_NODES does not match the library's node tile")
    x = _need(x, torch.float32, "node_features")
    w_edge = _edge_dtype(w_edge, "w_edge")
    sh_sorted = _need(sh_sorted, torch.float32, "sh_sorted")
    # destination rows = segments of rowptr (virtual nodes when the segments were cut by ops.csr_split); x rows are sources
    N, d_in = rowptr.shape[0] - 1, x.shape[1]
    if num_neigh is not None:
        num_neigh = _need(num_neigh, torch.float32, "num_neigh")
    agg = torch.empty(N, d_mid, dtype=torch.float32, device=x.device)
    with _timed(f"tp_scatter/d_mid={d_mid}"):
        rc = lib.matten_tp_paths(_ptr(x), d_in, _ptr(w_edge), w_edge.shape[1], _ptr(sh_sorted), sh_sorted.shape[1],
                                 _ptr(rowptr), _ptr(src_sorted), N, _ptr(entries), _ptr(unit_start),
                                 entries.shape[0], units_per_tile, d_mid, float(avg_num_neighbors or 0.0),
                                 _ptr(num_neigh), _ptr(agg), int(w_edge.dtype == torch.bfloat16), _stream())
    _lib.check(rc, "matten_tp_paths")
    return agg


def split_hidden(h2p: torch.Tensor) -> torch.Tensor:
    """fp32 [E,32] (already in the permuted column order) -> the split fp16 form [E,2,32] matten_tp_fused consumes:
    v = hi + 2^-11 lo.  Test/tool helper: the production path gets this form from matten_radial_hidden."""
    tiny = 2.0 ** -14
    hi = torch.where(h2p.abs() < tiny, torch.zeros_like(h2p), h2p.half().float())
    lo = (h2p - hi) * 2048.0
    lo = torch.where(lo.abs() < tiny, torch.zeros_like(lo), lo)
    return torch.stack([hi.half(), lo.half()], dim=1).contiguous()


def radial_hidden(geom_sorted, n_basis: int, r_start: float, r_end: float, w0p, w1p, h_scale=None) -> torch.Tensor:
    """-> h2s [E,2,32] fp16 (hi | lo pieces of the 32 hidden features, see include/matten_hip.h).
    h_scale: optional fp32 device tensor whose first element is the power-of-two output scale (RadialMLP.h_scale)"""
    lib = _lib.load()
    geom_sorted = _need(geom_sorted, torch.float32, "geom_sorted")
    w0p, w1p = _need(w0p, torch.float32, "w0p"), _need(w1p, torch.float32, "w1p")
    E = geom_sorted.shape[0]
    h2p = torch.empty(E, 2, 32, dtype=torch.float16, device=geom_sorted.device)
    with _timed("radial_hidden"):
        if h_scale is not None:
            h_scale = _need(h_scale, torch.float32, "h_scale")
        rc = lib.matten_radial_hidden(_ptr(geom_sorted), E, n_basis, r_start, r_end, _ptr(w0p), w0p.shape[0],
                                      _ptr(w1p), w0p.shape[1], _ptr(h2p), _ptr(h_scale), _stream())
    _lib.check(rc, "matten_radial_hidden")
    return h2p


def radial_hidden_multi(geom_sorted, n_basis: int, r_start: float, r_end: float, w0ps, w1ps, h_scales=None):
    """h2s of several radial MLPs over the same edges in one launch -> list of [E,2,32] fp16"""
    import ctypes

    lib = _lib.load()
    geom_sorted = _need(geom_sorted, torch.float32, "geom_sorted")
    w0ps = [_need(w, torch.float32, "w0p") for w in w0ps]
    w1ps = [_need(w, torch.float32, "w1p") for w in w1ps]
    L, E = len(w0ps), geom_sorted.shape[0]
    if not 1 <= L <= 8 or len(w1ps) != L or any(w.shape != w0ps[0].shape for w in w0ps):
        raise ValueError("radial_hidden_multi: 1..8 layers with equal basis / hidden sizes")
    out = [torch.empty(E, 2, 32, dtype=torch.float16, device=geom_sorted.device) for _ in range(L)]
    arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
    if h_scales is not None:
        h_scales = [_need(h, torch.float32, "h_scale") for h in h_scales]
        if len(h_scales) != L:
            raise ValueError("one h_scale per layer")
    with _timed("radial_hidden_multi"):
        rc = lib.matten_radial_hidden_multi(_ptr(geom_sorted), E, n_basis, r_start, r_end, arr(w0ps), w0ps[0].shape[0],
                                            arr(w1ps), w0ps[0].shape[1], arr(out),
                                            arr(h_scales) if h_scales is not None else None, L, _stream())
    _lib.check(rc, "matten_radial_hidden_multi")
    return out


def split_a_tiles(w2p: torch.Tensor, group_entries) -> Tuple[torch.Tensor, torch.Tensor]:
    """w2p [32, w_pad] -> (a_split [n_tiles, 64, 16] fp16, a_scale_inv [n_entries] fp32): the last radial layer as the
    MFMA A fragments matten_tp_fused consumes (layout: include/matten_hip.h), per entry scaled by the power of two
    that puts its largest magnitude in [2^13, 2^14) and split v = hi + 2^-11 lo like ops.split_hidden.  Runs on the
    device without a host sync; cached by the caller until the weights change."""
    dev = w2p.device
    g = torch.arange(4, device=dev)[:, None]
    kk = torch.arange(8, device=dev)[None, :]
    k_idx = 16 * (kk >> 2) + 4 * g + (kk & 3)                       # [4 (g), 8 (kk)]
    tiny = 2.0 ** -14
    tiles, inv = [], []
    for row in np.asarray(group_entries).reshape(-1, 32):
        w_base, n_mt = int(row[5]), int(row[7])
        if int(row[0]) < 0 or n_mt == 0:   # second-half record of a merged entry (plan.TP_KIND_MERGED): no tiles of its own
            inv.append(torch.ones((), device=dev))
            continue
        A = w2p[k_idx][:, :, w_base:w_base + 16 * n_mt]              # [g, kk, mt*16 + c]
        A = A.reshape(4, 8, n_mt, 16).permute(2, 0, 3, 1)            # [mt, g, c, kk]
        amax = A.abs().max()
        e = torch.where(amax > 0, (torch.frexp(amax)[1] - 1).clamp(-100, 100), torch.full_like(amax, 13, dtype=torch.int32))
        scale = torch.ldexp(torch.ones_like(amax), 13 - e)
        inv.append(torch.ldexp(torch.ones_like(amax), e - 13))
        v = A * scale
        hi = torch.where(v.abs() < tiny, torch.zeros_like(v), v.half().float())
        lo = (v - hi) * 2048.0
        lo = torch.where(lo.abs() < tiny, torch.zeros_like(lo), lo)
        tiles.append(torch.cat([hi.half(), lo.half()], dim=-1).reshape(n_mt, 64, 16))
    return torch.cat(tiles).contiguous(), torch.stack(inv).float().contiguous()


_GROUPS_CHECKED = False


def tp_fused(x, h2p, w2p, sh_sorted, rowptr, src_sorted, entries, unit_map, units_per_tile: int,
             lds_floats_per_wave: int, d_mid: int, avg_num_neighbors: float, num_neigh=None,
             a_split=None) -> torch.Tensor:
    """a_split: optional (fragments, scale_inv) from split_a_tiles(w2p, plan.group_entries)"""
    lib = _lib.load()
    from .plan import TP_TILE_NODES

    if lib.matten_tp_tile_nodes() != TP_TILE_NODES:
        raise _lib.MattenHipError("plan.TP_TILE_NODES does not match the library's node tile")
    from .plan import TP_MAX_COLS, TP_MAX_COLS_L0, TP_MAX_COLS_L1
    if (lib.matten_tp_max_cols() != TP_MAX_COLS or lib.matten_tp_max_cols_l0() != TP_MAX_COLS_L0
            or lib.matten_tp_max_cols_l1() != TP_MAX_COLS_L1):
        raise _lib.MattenHipError("plan.TP_MAX_COLS does not match the library's entry width (-DTPF_MAX_COLS)")
    global _GROUPS_CHECKED
    from .plan import tp_groups_hash
    if not _GROUPS_CHECKED and lib.matten_tp_groups_hash() != tp_groups_hash():
        raise _lib.MattenHipError("the library's coupling code (cg_gen.h) was generated for other coupling groups than plan.TP_GROUPS "
                                  "(MATTEN_TP_GROUPS / an interrupted tools/*.sh A/B build?): make -C matten_amd/csrc clean all")
    _GROUPS_CHECKED = True   # (one library per process: checked at the first call)
    x = _need_rows(x, torch.float32, "node_features")  # a column slice is fine: d_in below is the row stride
    h2p = _need(h2p, torch.float16, "h2s")
    if h2p.dim() != 3 or h2p.shape[1:] != (2, 32):
        raise ValueError(f"h2s must be [E,2,32] fp16 (ops.split_hidden / ops.radial_hidden), got {tuple(h2p.shape)}")
    w2p = _need(w2p, torch.float32, "w2p")
    sh_sorted = _need(sh_sorted, torch.float32, "sh_sorted")
    # destination rows = segments of rowptr (virtual nodes when the segments were cut by ops.csr_split); x rows are sources
    N, d_in = rowptr.shape[0] - 1, x.stride(0)
    if num_neigh is not None:
        num_neigh = _need(num_neigh, torch.float32, "num_neigh")
    if a_split is not None:
        a_split = (_need(a_split[0], torch.float16, "a_split"), _need(a_split[1], torch.float32, "a_scale_inv"))
        if a_split[1].numel() != entries.shape[0]:
            raise ValueError("a_scale_inv needs one value per group entry")
    if unit_map.numel() != units_per_tile:
        raise ValueError(f"unit_map has {unit_map.numel()} units, expected {units_per_tile} (plan.fused_unit_map)")
    agg = torch.empty(N, d_mid, dtype=torch.float32, device=x.device)
    with _timed(f"tp_scatter/d_mid={d_mid}/d_in={x.shape[1]}"):
        rc = lib.matten_tp_fused(_ptr(x), d_in, _ptr(h2p), _ptr(w2p), w2p.shape[1], _ptr(sh_sorted),
                                 sh_sorted.shape[1], _ptr(rowptr), _ptr(src_sorted), N, _ptr(entries),
                                 _ptr(unit_map), entries.shape[0], units_per_tile, lds_floats_per_wave, d_mid,
                                 float(avg_num_neighbors or 0.0), _ptr(num_neigh),
                                 _ptr(a_split[0]) if a_split is not None else None,
                                 _ptr(a_split[1]) if a_split is not None else None, _ptr(agg), _stream())
    _lib.check(rc, "matten_tp_fused")
    return agg


def agg_linear(agg, species_order, wtab, io_table, blocks, d_out: int, add=None) -> torch.Tensor:
    """out[N, d_out] = add + lin2(agg) for component-major neighbour sums (include/matten_hip.h matten_agg_linear;
    tables from plan.plan_agg_linear).  species_order: (order, seg) or None for a single species."""
    lib = _lib.load()
    from .plan import AGG_BLOCK, AGG_MAX_MT
    if lib.matten_agg_linear_block_chunks() != AGG_BLOCK or lib.matten_agg_linear_max_mt() != AGG_MAX_MT:
        raise _lib.MattenHipError("plan.AGG_BLOCK / AGG_MAX_MT do not match the library (-DAL_BLK_CHUNKS)")
    agg = _need(agg, torch.float32, "agg")
    wtab = _need(wtab, torch.float32, "A fragments")
    n_rows, ld = agg.shape
    order, seg = species_order if species_order is not None else (None, None)
    n_species = wtab.shape[0] if wtab.dim() == 2 else 1
    if add is not None:
        add = _need_rows(add, torch.float32, "add")
    out = torch.empty(n_rows, d_out, dtype=torch.float32, device=agg.device)
    with _timed(f"agg_linear/ld={ld}"):
        rc = lib.matten_agg_linear(_ptr(agg), ld, _ptr(order), _ptr(seg), n_species, _ptr(wtab), wtab.shape[-1],
                                   _ptr(io_table), io_table.shape[0], _ptr(blocks), blocks.shape[0], _ptr(add),
                                   add.stride(0) if add is not None else d_out, d_out, n_rows, _ptr(out), _stream())
    _lib.check(rc, "matten_agg_linear")
    return out


def agg_linear_gate(agg, species_order, wtab, io_table, blocks, d_out: int, cmeta, act_cst, d_act: int, add=None,
                    bn_scale=None, bn_shift=None) -> torch.Tensor:
    """agg_linear + the layer's Gate (+ eval-mode BatchNorm as per-column scale / shift) in one launch ->
    the activated row [N, d_act] (include/matten_hip.h matten_agg_linear_gate; cmeta from plan.plan_agg_gate)"""
    lib = _lib.load()
    from .plan import AGG_BLOCK, AGG_GATE_SETS, AGG_MAX_MT
    if (lib.matten_agg_linear_block_chunks() != AGG_BLOCK or lib.matten_agg_linear_max_mt() != AGG_MAX_MT
            or lib.matten_agg_linear_gate_sets() != AGG_GATE_SETS):
        raise _lib.MattenHipError("plan.AGG_BLOCK / AGG_MAX_MT / AGG_GATE_SETS do not match the library")
    agg = _need(agg, torch.float32, "agg")
    wtab = _need(wtab, torch.float32, "A fragments")
    n_rows, ld = agg.shape
    order, seg = species_order if species_order is not None else (None, None)
    n_species = wtab.shape[0] if wtab.dim() == 2 else 1
    if add is not None:
        add = _need_rows(add, torch.float32, "add")
    if cmeta.shape != (d_out, 4):
        raise ValueError("cmeta must be [d_out, 4]")
    if bn_scale is not None:
        bn_scale, bn_shift = _need(bn_scale, torch.float32, "bn_scale"), _need(bn_shift, torch.float32, "bn_shift")
        if bn_scale.numel() != d_act or bn_shift.numel() != d_act:
            raise ValueError("bn_scale / bn_shift must have one value per activated column")
    out = torch.empty(n_rows, d_act, dtype=torch.float32, device=agg.device)
    with _timed(f"agg_linear/ld={ld}"):
        rc = lib.matten_agg_linear_gate(_ptr(agg), ld, _ptr(order), _ptr(seg), n_species, _ptr(wtab), wtab.shape[-1],
                                        _ptr(io_table), io_table.shape[0], _ptr(blocks), blocks.shape[0], _ptr(add),
                                        add.stride(0) if add is not None else d_out, d_out, n_rows, _ptr(cmeta),
                                        _ptr(act_cst), _ptr(bn_scale), _ptr(bn_shift), d_act, _ptr(out), _stream())
    _lib.check(rc, "matten_agg_linear_gate")
    return out


_SL_ROWS_LDS_BYTES = 52 * 1024  # three workgroups of matten_species_linear_rows per CU


def species_linear(x, species_order, wp, w_stride: int, item_tables, d_out: int, add=None,
                   fully_covered: bool = True, variant: Optional[str] = None) -> torch.Tensor:
    """species_order: None (plain linear) or (order[N] i32, seg[S+1] i32) = nodes sorted by species.
    item_tables: list of int32 [n_items,8] tensors (passes).  out = add + sum_passes.
    variant: None = choose (row-resident kernel when rows + weights fit LDS), "rows" / "stream" force one (self-check)."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    from . import selfcheck
    selfcheck.check_species_linear(x.device)   # first call per device: both inline-asm MFMA kernels on exact integer cases
    wp = _need(wp, torch.float32, "packed weights")
    n_rows, d_in = x.shape
    order, seg = species_order if species_order is not None else (None, None)
    n_species = wp.shape[0] if wp.dim() == 2 else 1
    cur_add = add
    if cur_add is not None:
        cur_add = _need_rows(cur_add, torch.float32, "add")
    if fully_covered:
        out = torch.empty(n_rows, d_out, dtype=torch.float32, device=x.device)
    else:  # irreps without an input path stay zero (e3nn output_mask semantics)
        out = cur_add.clone() if cur_add is not None else torch.zeros(n_rows, d_out, dtype=torch.float32, device=x.device)
    # short rows (node features: lin1 / self-connection, first-layer lin2, read-out): rows and weights resident in LDS
    rows_fits = len(item_tables) == 1 and 4 * (16 * (d_in | 1) + w_stride + 8 * item_tables[0].shape[0] + 4) <= _SL_ROWS_LDS_BYTES
    rows_variant = rows_fits and os.environ.get("MATTEN_SL_ROWS", "1") != "0"
    if variant is not None:
        rows_variant = rows_fits and variant == "rows"
    for items in item_tables:
        fn = lib.matten_species_linear_rows if rows_variant else lib.matten_species_linear
        _lib.check(
            fn(_ptr(x), d_in, _ptr(order), _ptr(seg), n_species, _ptr(wp), w_stride,
                                      _ptr(items), items.shape[0], d_out, _ptr(cur_add),
                                      cur_add.stride(0) if cur_add is not None else d_out, n_rows, _ptr(out), _stream()),
            "matten_species_linear",
        )
        cur_add = out
    return out


def gate_bn(x, meta, act_cst, running_mean=None, running_var=None, bn_weight=None, bn_bias=None, eps: float = 1e-5):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    n_rows, d_in = x.shape
    d_out = meta.shape[0]
    out = torch.empty(n_rows, d_out, dtype=torch.float32, device=x.device)
    _lib.check(
        lib.matten_gate_bn(_ptr(x), d_in, _ptr(meta), d_out, _ptr(act_cst), _ptr(running_mean), _ptr(running_var),
                           _ptr(bn_weight), _ptr(bn_bias), eps, n_rows, _ptr(out), _stream()),
        "matten_gate_bn",
    )
    return out


def segment_reduce(x, ptr, mean: bool) -> torch.Tensor:
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    ptr = _need(ptr, torch.int64, "ptr")
    B = ptr.shape[0] - 1
    out = torch.empty(B, x.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(lib.matten_segment_reduce(_ptr(x), x.shape[1], _ptr(ptr), B, int(mean), _ptr(out), _stream()),
               "matten_segment_reduce")
    return out


def segment_minmax(x, ptr, take_max: bool, want_arg: bool = False):
    """per crystal and column min / max -> out [B, dim] (and the source rows [B, dim] int64 when want_arg)"""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    ptr = _need(ptr, torch.int64, "ptr")
    B = ptr.shape[0] - 1
    out = torch.empty(B, x.shape[1], dtype=torch.float32, device=x.device)
    arg = torch.empty(B, x.shape[1], dtype=torch.int64, device=x.device) if want_arg else None
    _lib.check(lib.matten_segment_minmax(_ptr(x), x.shape[1], _ptr(ptr), B, int(take_max), _ptr(out), _ptr(arg), _stream()),
               "matten_segment_minmax")
    return (out, arg) if want_arg else out


def segment_minmax_bwd(dy, arg, n_rows: int) -> torch.Tensor:
    lib = _lib.load()
    dy = _need(dy, torch.float32, "dy")
    dx = torch.zeros(n_rows, dy.shape[1], dtype=torch.float32, device=dy.device)
    _lib.check(lib.matten_segment_minmax_bwd(_ptr(dy), dy.shape[1], _ptr(arg), dy.shape[0], _ptr(dx), _stream()),
               "matten_segment_minmax_bwd")
    return dx


def dense_rows(x, q) -> torch.Tensor:
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    q = _need(q, torch.float32, "q")
    out = torch.empty(x.shape[0], q.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(lib.matten_dense_rows(_ptr(x), x.shape[1], _ptr(q), q.shape[1], x.shape[0], _ptr(out), _stream()),
               "matten_dense_rows")
    return out


# ---------------------------------------------------------------------------------------------------
# adjoint operators (training step)
# ---------------------------------------------------------------------------------------------------
def tp_backward(x, w_edge, sh_sorted, src_sorted, dst_sorted, col_meta, nnz_ijk, nnz_c, g_agg,
                avg_num_neighbors: float, num_neigh=None, in_groups=None):
    """-> (dx [N,d_in], dw [E,W]) for agg = tp(x, w_edge) with w_edge in the reference column order.
    in_groups: optional (in_ptr [n_in+1] i32, in_cols [W] i32) = the columns grouped by input channel (plan.bw_in_*)."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    w_edge = _edge_dtype(w_edge, "w_edge")
    g_agg = _need(g_agg, torch.float32, "grad agg")
    N, d_in = x.shape
    E = w_edge.shape[0]
    W = col_meta.shape[0]
    dx = torch.zeros(N, d_in, dtype=torch.float32, device=x.device)
    dw = torch.empty(E, w_edge.shape[1], dtype=w_edge.dtype, device=x.device)   # same row stride and storage as w_edge
    if num_neigh is not None:
        num_neigh = _need(num_neigh, torch.float32, "num_neigh")
    with _timed(f"tp_backward/d_mid={g_agg.shape[1]}/d_in={d_in}"):
        rc = lib.matten_tp_backward(_ptr(x), d_in, _ptr(w_edge), w_edge.shape[1], _ptr(sh_sorted), sh_sorted.shape[1],
                                    _ptr(src_sorted), _ptr(dst_sorted), _ptr(col_meta), W, _ptr(nnz_ijk), _ptr(nnz_c),
                                    _ptr(g_agg), g_agg.shape[1], float(avg_num_neighbors or 0.0), _ptr(num_neigh), E,
                                    _ptr(dx), _ptr(dw), dw.shape[1], _ptr(in_groups[0]) if in_groups else None,
                                    _ptr(in_groups[1]) if in_groups else None,
                                    in_groups[0].shape[0] - 1 if in_groups else 0,
                                    int(w_edge.dtype == torch.bfloat16), _stream())
    _lib.check(rc, "matten_tp_backward")
    return dx, dw


def split_a_tiles_dev(w2p, entries, n_tiles: int, h_scale=None):
    """-> (frag [n_tiles, 64, 16] fp16, scale_inv [n_entries]): matten_split_a_tiles on the device, no host read (the values of
    split_a_tiles; h_scale = the [s, 1/s] pair of matten_radial_h_scale folds 1/s into scale_inv)"""
    lib = _lib.load()
    w2p = _need(w2p, torch.float32, "w2p")
    entries = _need(entries, torch.int32, "entries")
    frag = torch.empty(n_tiles, 64, 16, dtype=torch.float16, device=w2p.device)
    inv = torch.empty(entries.shape[0], dtype=torch.float32, device=w2p.device)
    _lib.check(lib.matten_split_a_tiles(_ptr(w2p), w2p.shape[1], _ptr(entries), entries.shape[0], _ptr(h_scale), _ptr(frag),
                                        _ptr(inv), _stream()), "matten_split_a_tiles")
    return frag, inv


WFREE_MAX_MUL = 256   # widest input block (channels) matten_tp_backward_lit_wfree takes: one workgroup = 256 / lanes-per-edge edges


def tp_backward_lit(x, w_edge, sh_sorted, src_sorted, dst_sorted, blocks, paths, sum_lanes: int, g_agg,
                    avg_num_neighbors: float, num_neigh=None, out_csr=None, blocks_cover_input: bool = True, wfree=None,
                    dw_shape=None, lds_floats: int = 4096, max_l: int = 4, max_mul: int = 256):
    """the adjoint of tp_backward with literal-coefficient coupling code (include/matten_hip.h matten_tp_backward_lit;
    tables plan.bw_blocks / bw_paths) -> (dx [N,d_in], dw [E, ld of w_edge]).
    out_csr = (out_ptr [N+1] i32, out_perm [E] i32): sorted-edge indices grouped by SOURCE node -- dx is then summed per
    node in that fixed order (bitwise reproducible) instead of through atomics.
    wfree = (h2s [E, 2, 32] fp16, frag, w_inv) with w_edge None: the weights are re-evaluated inside the kernel on the matrix
    cores (matten_tp_backward_lit_wfree; frag / w_inv from split_a_tiles_dev over plan.bw_w_entries); dw_shape = ((E, ld), dtype);
    max_mul = plan.bw_max_mul, the widest block's channel count (the library refuses > 256: WFREE_MAX_MUL)."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    g_agg = _need(g_agg, torch.float32, "grad agg")
    N, d_in = x.shape
    if wfree is not None:
        if w_edge is not None or dw_shape is None:
            raise ValueError("wfree: pass w_edge=None and dw_shape=((E, ld), dtype)")
        h2s, frag, w_inv = (_need(t, dt, n) for t, dt, n in zip(wfree, (torch.float16, torch.float16, torch.float32),
                                                                ("h2s", "frag", "w_inv")))
        (E, dw_ld), dw_dtype = dw_shape
        if h2s.shape != (E, 2, 32) or w_inv.numel() != paths.shape[0]:
            raise ValueError("wfree: h2s must be [E, 2, 32] and w_inv hold one value per path")
    else:
        w_edge = _edge_dtype(w_edge, "w_edge")
        E, dw_ld, dw_dtype = w_edge.shape[0], w_edge.shape[1], w_edge.dtype
    if out_csr is not None:
        out_ptr, out_perm = (_need(t, torch.int32, n) for t, n in zip(out_csr, ("out_ptr", "out_perm")))
        dx = torch.empty(N, d_in, dtype=torch.float32, device=x.device)
        dx_edges = (torch.empty if blocks_cover_input else torch.zeros)(E, d_in, dtype=torch.float32, device=x.device)
    else:
        out_ptr = out_perm = dx_edges = None
        dx = torch.zeros(N, d_in, dtype=torch.float32, device=x.device)
    # columns no path writes (the pad up to the row stride) must not hold NaN for the MLP adjoint's masked reads: they are
    # masked by a select there, so plain empty storage is fine
    dw = torch.empty(E, dw_ld, dtype=dw_dtype, device=x.device)
    if num_neigh is not None:
        num_neigh = _need(num_neigh, torch.float32, "num_neigh")
    if wfree is not None:
        with _timed(f"tp_backward/d_mid={g_agg.shape[1]}/d_in={d_in}"):
            rc = lib.matten_tp_backward_lit_wfree(_ptr(x), d_in, _ptr(h2s), _ptr(frag), _ptr(w_inv), _ptr(sh_sorted),
                                                  sh_sorted.shape[1], _ptr(src_sorted), _ptr(dst_sorted), _ptr(blocks),
                                                  blocks.shape[0], int(sum_lanes), _ptr(paths), paths.shape[0], _ptr(g_agg),
                                                  g_agg.shape[1], float(avg_num_neighbors or 0.0), _ptr(num_neigh), E, _ptr(dx),
                                                  _ptr(dw), dw_ld, int(dw_dtype == torch.bfloat16), N, _ptr(out_ptr),
                                                  _ptr(out_perm), _ptr(dx_edges), int(lds_floats), int(max_l), int(max_mul),
                                                  _stream())
        _lib.check(rc, "matten_tp_backward_lit_wfree")
        return dx, dw
    with _timed(f"tp_backward/d_mid={g_agg.shape[1]}/d_in={d_in}"):
        rc = lib.matten_tp_backward_lit(_ptr(x), d_in, _ptr(w_edge), w_edge.shape[1], _ptr(sh_sorted), sh_sorted.shape[1],
                                        _ptr(src_sorted), _ptr(dst_sorted), _ptr(blocks), blocks.shape[0], int(sum_lanes),
                                        _ptr(paths), paths.shape[0], _ptr(g_agg), g_agg.shape[1],
                                        float(avg_num_neighbors or 0.0), _ptr(num_neigh), E, _ptr(dx), _ptr(dw),
                                        dw.shape[1], int(w_edge.dtype == torch.bfloat16), N, _ptr(out_ptr), _ptr(out_perm),
                                        _ptr(dx_edges), int(max_l), _stream())
    _lib.check(rc, "matten_tp_backward_lit")
    return dx, dw


def species_linear_wgrad(x, dy, species_order, n_species: int, seg_tables, w_stride: int) -> torch.Tensor:
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    dy = _need(dy, torch.float32, "dy")
    order, seg = species_order if species_order is not None else (None, None)
    # every packed weight is written exactly once (nothing to pre-zero); large batches cut the species' rows into slices
    # (a compact list of (species, slice) items) whose partial sums the library adds in slice order
    n_scratch = lib.matten_species_linear_wgrad_scratch_floats(x.shape[0], n_species, w_stride)
    dwp = torch.empty(n_species, w_stride, dtype=torch.float32, device=x.device)
    partial = torch.empty(n_scratch, dtype=torch.float32, device=x.device) if n_scratch > 0 else None
    for segs in seg_tables:
        _lib.check(
            lib.matten_species_linear_wgrad(_ptr(x), x.shape[1], _ptr(dy), dy.shape[1], _ptr(order), _ptr(seg),
                                            n_species, x.shape[0], _ptr(segs), segs.shape[0], w_stride, _ptr(dwp),
                                            _ptr(partial), _stream()),
            "matten_species_linear_wgrad",
        )
    return dwp


def gate_bwd(x, meta, act_cst, dy) -> torch.Tensor:
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    dy = _need(dy, torch.float32, "dy")
    dx = torch.empty_like(x)   # every input column (scalars, gates, gated) is written exactly once
    _lib.check(lib.matten_gate_bwd(_ptr(x), x.shape[1], _ptr(meta), meta.shape[0], _ptr(act_cst), _ptr(dy), x.shape[0],
                                   _ptr(dx), _stream()), "matten_gate_bwd")
    return dx


def _bn_scratch(lib, x):
    """partial records of the two-stage whole-batch reductions (large batches only)"""
    n = lib.matten_bn_scratch_floats(x.shape[0], x.shape[1])
    return torch.empty(n, dtype=torch.float32, device=x.device) if n > 0 else None


def bn_train_fwd(x, col2chan, chan, weight, bias, eps: float, running_mean=None, running_var=None, momentum: float = 0.0):
    """batch statistics -> (y, mean, nu); running_mean / running_var (optional) are updated in place by the same launch"""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    C = chan.shape[0]
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    nu = torch.empty(C, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    if running_var is not None:
        running_mean = _need(running_mean, torch.float32, "running_mean")
        running_var = _need(running_var, torch.float32, "running_var")
    scratch = _bn_scratch(lib, x)
    _lib.check(lib.matten_bn_train_fwd(_ptr(x), x.shape[1], x.shape[0], _ptr(col2chan), _ptr(chan), C, _ptr(weight),
                                       _ptr(bias), eps, _ptr(mean), _ptr(nu), _ptr(y), _ptr(running_mean),
                                       _ptr(running_var), float(momentum), _ptr(scratch), _stream()),
               "matten_bn_train_fwd")
    if running_var is not None:
        bump_weights_epoch()   # the running statistics were updated through raw pointers (caches of the folded BatchNorm)
    return y, mean, nu


def norm_act(x, chan, act: int, epsilon: float = 1e-8, running_mean=None, running_var=None, bn_weight=None, bn_bias=None,
             bn_eps: float = 1e-5) -> torch.Tensor:
    """e3nn NormActivation (+ optional eval-mode BatchNorm), include/matten_hip.h matten_norm_act"""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    y = torch.empty_like(x)
    _lib.check(lib.matten_norm_act(_ptr(x), x.shape[1], x.shape[0], _ptr(chan), chan.shape[0], int(act), float(epsilon),
                                   _ptr(running_mean), _ptr(running_var), _ptr(bn_weight), _ptr(bn_bias), float(bn_eps),
                                   _ptr(y), _stream()), "matten_norm_act")
    return y


def norm_act_bwd(x, dy, chan, act: int, epsilon: float = 1e-8) -> torch.Tensor:
    lib = _lib.load()
    dy = _need(dy, torch.float32, "dy")
    dx = torch.empty_like(x)
    _lib.check(lib.matten_norm_act_bwd(_ptr(x), _ptr(dy), x.shape[1], x.shape[0], _ptr(chan), chan.shape[0], int(act),
                                       float(epsilon), _ptr(dx), _stream()), "matten_norm_act_bwd")
    return dx


def instance_norm_fwd(x, seg_ptr, seg_of_row, col2chan, chan, weight, bias, eps: float):
    """per-crystal statistics (reference InstanceNorm, nn/utils.py:448-588) -> (y, mean [B, C], nu [B, C])"""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    seg_ptr = _need(seg_ptr, torch.int64, "ptr")
    seg_of_row = _need(seg_of_row, torch.int64, "batch")
    C, B = chan.shape[0], seg_ptr.shape[0] - 1
    mean = torch.empty(B, C, dtype=torch.float32, device=x.device)
    nu = torch.empty(B, C, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    _lib.check(lib.matten_instance_norm_fwd(_ptr(x), x.shape[1], x.shape[0], _ptr(seg_ptr), _ptr(seg_of_row), B,
                                            _ptr(col2chan), _ptr(chan), C, _ptr(weight), _ptr(bias), eps, _ptr(mean),
                                            _ptr(nu), _ptr(y), _stream()), "matten_instance_norm_fwd")
    return y, mean, nu


def instance_norm_bwd(x, dy, seg_ptr, seg_of_row, col2chan, chan, mean, nu, weight, eps: float):
    lib = _lib.load()
    dy = _need(dy, torch.float32, "dy")
    B, C = mean.shape
    A = torch.empty(B, C, dtype=torch.float32, device=x.device)
    Bs = torch.empty(B, C, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    _lib.check(lib.matten_instance_norm_bwd(_ptr(x), _ptr(dy), x.shape[1], x.shape[0], _ptr(seg_ptr), _ptr(seg_of_row), B,
                                            _ptr(col2chan), _ptr(chan), C, _ptr(mean), _ptr(nu), _ptr(weight), eps,
                                            _ptr(A), _ptr(Bs), _ptr(dx), _stream()), "matten_instance_norm_bwd")
    return dx, A, Bs


def bn_train_bwd(x, dy, col2chan, chan, mean, nu, weight, eps: float, n_bias: int):
    """-> (dx, dweight [C], dbias [n_bias]): the parameter gradients come out of the reduction kernel itself"""
    lib = _lib.load()
    dy = _need(dy, torch.float32, "dy")
    C = chan.shape[0]
    A = torch.empty(C, dtype=torch.float32, device=x.device)
    B = torch.empty(C, dtype=torch.float32, device=x.device)
    dweight = torch.empty(C, dtype=torch.float32, device=x.device)
    dbias = torch.empty(n_bias, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    scratch = _bn_scratch(lib, x)
    _lib.check(lib.matten_bn_train_bwd(_ptr(x), _ptr(dy), x.shape[1], x.shape[0], _ptr(col2chan), _ptr(chan), C,
                                       _ptr(mean), _ptr(nu), _ptr(weight), eps, _ptr(A), _ptr(B), _ptr(dx), _ptr(dweight),
                                       _ptr(dbias), _ptr(scratch), _stream()),
               "matten_bn_train_bwd")
    return dx, dweight, dbias


def segment_reduce_bwd(dy, ptr, n_rows: int, mean: bool) -> torch.Tensor:
    lib = _lib.load()
    dy = _need(dy, torch.float32, "dy")
    ptr = _need(ptr, torch.int64, "ptr")
    dx = torch.zeros(n_rows, dy.shape[1], dtype=torch.float32, device=dy.device)
    _lib.check(lib.matten_segment_reduce_bwd(_ptr(dy), dy.shape[1], _ptr(ptr), dy.shape[0], int(mean), _ptr(dx),
                                             _stream()), "matten_segment_reduce_bwd")
    return dx


def graph_prep(pos64, cell64, ptr, r_cut: float):
    """per-crystal prologue of the device graph builder (include/matten_hip.h matten_graph_prep)
    -> (frac [N,3] f64, bound [B,3] f64, batch [N] i64, pos [N,3] f32, cell [3B,3] f32)"""
    lib = _lib.load()
    pos64 = _need(pos64, torch.float64, "pos")
    cell64 = _need(cell64, torch.float64, "cell")
    ptr = _need(ptr, torch.int64, "ptr")
    N, B, dev = pos64.shape[0], ptr.shape[0] - 1, pos64.device
    frac = torch.empty(N, 3, dtype=torch.float64, device=dev)
    bound = torch.empty(B, 3, dtype=torch.float64, device=dev)
    batch = torch.empty(N, dtype=torch.int64, device=dev)
    pos32 = torch.empty(N, 3, dtype=torch.float32, device=dev)
    cell32 = torch.empty(3 * B, 3, dtype=torch.float32, device=dev)
    _lib.check(lib.matten_graph_prep(_ptr(pos64), _ptr(cell64), _ptr(ptr), B, float(r_cut), _ptr(frac), _ptr(bound),
                                     _ptr(batch), _ptr(pos32), _ptr(cell32), _stream()), "matten_graph_prep")
    return frac, bound, batch, pos32, cell32


def neighbor_list(pos64, cell64, ptr, frac, bound, pair_ptr, r_cut: float, max_atoms: int, n_pairs: int, want_csr: bool = True):
    """Periodic neighbour list of a batch of crystals, canonical (i, j, Sx, Sy, Sz) order.
    frac / bound: ops.graph_prep; pair_ptr[B+1] = running sum of n_b^2 (ordered pairs numbered crystal by crystal,
    i-major), n_pairs = pair_ptr[B].
    -> (edge_index [2,E] i64 (global ids), edge_cell_shift [E,3] f32, num_neigh [N] f32 (edges per centre atom),
        pair_offsets [n_pairs+1] i64, smallest edge count of a crystal: an int, read back together with the edge count,
        csr = (perm [E] i32, rowptr [N+1] i32, src_sorted [E] i32) -- the destination-sorted view ops.csr_build would derive
        from edge_index, bit for bit -- or None)"""
    lib = _lib.load()
    pos64 = _need(pos64, torch.float64, "pos")
    cell64 = _need(cell64, torch.float64, "cell")
    ptr = _need(ptr, torch.int64, "ptr")
    frac = _need(frac, torch.float64, "frac")
    bound = _need(bound, torch.float64, "bound")
    pair_ptr = _need(pair_ptr, torch.int64, "pair_ptr")
    B = ptr.shape[0] - 1
    N = pos64.shape[0]
    dev = pos64.device
    # i-major counts in the first half, the same counts numbered j-major in the second: ONE scan serves both orders
    counts = torch.empty((2 if want_csr else 1) * n_pairs, dtype=torch.int32, device=dev)
    with _timed("neighbor_count"):
        _lib.check(
            lib.matten_neighbor_count(_ptr(pos64), _ptr(cell64), _ptr(ptr), _ptr(frac), _ptr(bound), _ptr(pair_ptr),
                                      float(r_cut), B, int(max_atoms), _ptr(counts),
                                      counts.data_ptr() + 4 * n_pairs if (want_csr and n_pairs) else None, _stream()),
            "matten_neighbor_count",
        )
    scan = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, dtype=torch.int64, out=scan[1:])
    offsets = scan[: n_pairs + 1]
    # the one host sync of graph construction: the edge count sizes the outputs; the smallest edge count of a crystal
    # (0: the caller has to find and report the edgeless ones) rides on the same read-back
    if n_pairs:
        summary = torch.empty(2, dtype=torch.int64, device=dev)
        _lib.check(lib.matten_neighbor_summary(_ptr(offsets), _ptr(pair_ptr), B, _ptr(summary), _stream()),
                   "matten_neighbor_summary")
        E, min_edges = summary.tolist()
    else:
        E, min_edges = 0, 0
    edge_index = torch.empty(2, E, dtype=torch.int64, device=dev)
    shifts = torch.empty(E, 3, dtype=torch.float32, device=dev)
    num_neigh = torch.empty(N, dtype=torch.float32, device=dev)
    csr = offsets_t = None
    if want_csr and n_pairs and E < 2 ** 31:
        offsets_t = scan[n_pairs:]               # the j-major half of the scan (starts at E: the kernel subtracts it)
        csr = (torch.empty(E, dtype=torch.int32, device=dev), torch.empty(N + 1, dtype=torch.int32, device=dev),
               torch.empty(E, dtype=torch.int32, device=dev))
    with _timed("neighbor_fill"):
        _lib.check(
            lib.matten_neighbor_fill(_ptr(pos64), _ptr(cell64), _ptr(ptr), _ptr(frac), _ptr(bound), _ptr(pair_ptr),
                                     float(r_cut), B, int(max_atoms), _ptr(offsets), E, _ptr(edge_index), _ptr(shifts),
                                     _ptr(num_neigh), _ptr(offsets_t), N, _ptr(csr[1]) if csr else None,
                                     _ptr(csr[2]) if csr else None, _ptr(csr[0]) if csr else None, _stream()),
            "matten_neighbor_fill",
        )
    return edge_index, shifts, num_neigh, offsets, int(min_edges), csr
