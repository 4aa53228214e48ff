"""
Host-side crystal graph construction and batching (the producer side of the backbone's data dict).

Contract kept from the reference (data/data.py:285-413, which delegates to ASE):
  * edges are all ordered triples (i, j, S) with |r_j + S.cell - r_i| < r_cut (strict, fp64),
    periodic in x, y, z, minus the true self edges (i == j and S == 0)
  * edge_index[0] = i (centre), edge_index[1] = j (neighbour); edge_cell_shift = S as float
  * num_neigh = bincount(i); cell rows are lattice vectors
ASE leaves the order within a centre atom unspecified; here edges come out in the canonical
lexicographic order (i, j, Sx, Sy, Sz).

``collate`` lays a list of crystals out as the flat struct-of-arrays batch that PyG's
``Batch.from_data_list`` + ``tensor_property_to_dict`` (data/data.py:146-159) would produce.
"""
from typing import Dict, List, Optional, Sequence

import os

import numpy as np
import torch
from scipy.spatial import cKDTree


def image_reach(pos: np.ndarray, cell: np.ndarray, r_cut: float) -> np.ndarray:
    """Number of periodic images to scan along each lattice direction so that no pair within r_cut is missed:
    cutoff over the lattice-plane spacing, plus the fractional extent of the atoms (they need not be wrapped)."""
    recip = np.linalg.inv(cell).T  # rows: reciprocal vectors (without 2 pi)
    plane_dist = 1.0 / np.linalg.norm(recip, axis=1)
    frac = pos @ np.linalg.inv(cell)
    span = frac.max(0) - frac.min(0) if len(pos) else np.zeros(3)
    return np.ceil(float(r_cut) / plane_dist + span).astype(int)


def neighbor_list(pos: np.ndarray, cell: np.ndarray, r_cut: float):
    """-> edge_index [2,E] int64, shifts [E,3] int64 (canonical order).  Host builder (scipy KD-tree); the
    production path for batches is ``batch_graphs_gpu`` below, which emits the identical list on the device."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    n = pos.shape[0]
    rc = float(r_cut)
    reach = image_reach(pos, cell, rc)

    # candidate images: every lattice shift in the reach box; a KD-tree over the image atoms prunes
    # the pair search, the strict fp64 test below decides (same expression as the brute-force form)
    rng = [np.arange(-m, m + 1) for m in reach]
    S = np.stack(np.meshgrid(*rng, indexing="ij"), axis=-1).reshape(-1, 3)
    T = S @ cell
    img = (pos[None, :, :] + T[:, None, :]).reshape(-1, 3)  # [ns*n, 3], index = s*n + j
    tree = cKDTree(img)
    cand = tree.query_ball_point(pos, rc * (1.0 + 1e-9) + 1e-9)
    i = np.repeat(np.arange(n), [len(c) for c in cand])
    flat = np.concatenate([np.asarray(c, dtype=np.int64) for c in cand]) if len(i) else np.zeros(0, dtype=np.int64)
    sidx, j = flat // n, flat % n
    d = (pos[j] + T[sidx]) - pos[i]
    ok = np.sqrt((d * d).sum(-1)) < rc
    ok &= ~((i == j) & np.all(S[sidx] == 0, axis=1))
    i, j, s = i[ok], j[ok], S[sidx[ok]]
    if i.size == 0:
        raise ValueError("After eliminating self edges, no edges remain in this system.")
    order = np.lexsort((s[:, 2], s[:, 1], s[:, 0], j, i))
    return np.stack([i[order], j[order]]).astype(np.int64), s[order].astype(np.int64)


def crystal_graph(pos, cell, atomic_numbers, r_cut: float, y: Optional[Dict[str, torch.Tensor]] = None,
                  **extra) -> Dict[str, torch.Tensor]:
    """One crystal as the tensors a reference ``Crystal`` data point carries (data/data.py:134-144)."""
    pos = np.asarray(pos, dtype=np.float64)
    cell = np.asarray(cell, dtype=np.float64)
    edge_index, shifts = neighbor_list(pos, cell, r_cut)
    g = {
        "pos": torch.as_tensor(pos, dtype=torch.float32),
        "edge_index": torch.as_tensor(edge_index),
        "edge_cell_shift": torch.as_tensor(shifts, dtype=torch.float32),
        "cell": torch.as_tensor(cell, dtype=torch.float32),
        "num_neigh": torch.as_tensor(np.bincount(edge_index[0], minlength=len(pos)), dtype=torch.float32),
        "atomic_numbers": torch.as_tensor(np.asarray(atomic_numbers, dtype=np.int64)),
    }
    for k, v in {**(y or {}), **extra}.items():
        g[k] = torch.as_tensor(v)
    return g


def collate(graphs: Sequence[Dict[str, torch.Tensor]], device=None, pin: bool = False) -> Dict[str, torch.Tensor]:
    """Disjoint union of crystals: node offsets added to edge_index, `batch` and `ptr` appended."""
    sizes = [int(g["pos"].shape[0]) for g in graphs]
    ptr = np.zeros(len(graphs) + 1, dtype=np.int64)
    np.cumsum(sizes, out=ptr[1:])
    out: Dict[str, torch.Tensor] = {}
    keys = list(graphs[0].keys())
    for k in keys:
        if k == "edge_index":
            out[k] = torch.cat([g[k] + int(o) for g, o in zip(graphs, ptr[:-1])], dim=1)
        else:
            out[k] = torch.cat([g[k] for g in graphs], dim=0)
    out["batch"] = torch.repeat_interleave(torch.arange(len(graphs), dtype=torch.int64), torch.as_tensor(sizes))
    out["ptr"] = torch.from_numpy(ptr)
    if pin:
        out = {k: v.pin_memory() for k, v in out.items()}
    if device is not None:
        out = {k: v.to(device, non_blocking=pin) for k, v in out.items()}
    return out


def average_num_neighbors(graphs: Sequence[Dict[str, torch.Tensor]]) -> float:
    """dataset statistic the reference derives in get_to_model_info (dataset/structure_scalar_tensor.py:640-666)."""
    return float(torch.cat([g["num_neigh"] for g in graphs]).mean())


_MAX_CRYSTALS_PER_LAUNCH = 65535
EMIT_CSR = os.environ.get("MATTEN_GRAPH_EMIT_CSR", "1") != "0"   # device builder: batches carry their destination-sorted CSR


class EdgelessStructures(ValueError):
    """Some crystals of a batch have no edge inside the cutoff; ``indices`` are their positions in the batch."""

    def __init__(self, indices):
        super().__init__(f"After eliminating self edges, no edges remain in this system (structures {list(indices)}).")
        self.indices = list(indices)


def batch_graphs_gpu(structures: Sequence, r_cut: float, device="cuda", y: Optional[Dict[str, torch.Tensor]] = None,
                     ) -> Dict[str, torch.Tensor]:
    """Crystals -> collated batch, with the neighbour search on the GPU (matten_neighbor_count/_fill).

    ``structures`` is a sequence of (pos [n,3], cell [3,3], atomic_numbers [n]) triples; they are packed into the
    flat struct-of-arrays form of ``batch_graphs_gpu_soa``, which callers that already hold their crystals as
    arrays should use directly (no per-structure Python work)."""
    sizes = np.array([len(s[0]) for s in structures], dtype=np.int64)
    ptr = np.zeros(len(structures) + 1, dtype=np.int64)
    np.cumsum(sizes, out=ptr[1:])
    pos = np.concatenate([np.asarray(s[0], dtype=np.float64).reshape(-1, 3) for s in structures])
    cell = np.stack([np.asarray(s[1], dtype=np.float64).reshape(3, 3) for s in structures])
    Z = np.concatenate([np.asarray(s[2], dtype=np.int64).reshape(-1) for s in structures])
    return batch_graphs_gpu_soa(pos, cell, Z, ptr, r_cut, device, y)


def batch_graphs_gpu_soa(pos: np.ndarray, cell: np.ndarray, Z: np.ndarray, ptr: np.ndarray, r_cut: float,
                         device="cuda", y: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Flat batch (pos [N,3] fp64, cell [B,3,3] fp64, Z [N] int64, ptr [B+1] int64) ->
    collated graph batch on the device.

    The result has exactly the keys, dtypes and edge order of ``collate([crystal_graph(...) ...], device)``; only
    positions, cells and species cross PCIe (fp64 for the distance test, as in the host builder).  A crystal without
    any edge raises ``EdgelessStructures`` (a ValueError, like the reference data/data.py:398-402)."""
    from .. import ops

    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    Z = np.ascontiguousarray(Z, dtype=np.int64).reshape(-1)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    sizes = np.diff(ptr)
    n_crystals = len(sizes)
    if n_crystals > _MAX_CRYSTALS_PER_LAUNCH:  # blockIdx.y of the neighbour kernels: build in slabs and concatenate
        parts = []
        for lo in range(0, n_crystals, _MAX_CRYSTALS_PER_LAUNCH):
            hi = min(n_crystals, lo + _MAX_CRYSTALS_PER_LAUNCH)
            a, b = ptr[lo], ptr[hi]
            try:
                parts.append(batch_graphs_gpu_soa(pos[a:b], cell[lo:hi], Z[a:b], ptr[lo : hi + 1] - a, r_cut, device))
                for k in [k for k in parts[-1] if k.startswith("_amd_")]:
                    del parts[-1][k]   # a slab's own CSR does not concatenate: the forward builds the whole batch's
            except EdgelessStructures as e:
                raise EdgelessStructures([lo + k for k in e.indices]) from None
        out, node_off, cry_off = {}, 0, 0
        for p in parts:
            p["edge_index"] = p["edge_index"] + node_off
            p["batch"] = p["batch"] + cry_off
            node_off += p["pos"].shape[0]
            cry_off += p["ptr"].shape[0] - 1
        for k in parts[0]:
            if k == "edge_index":
                out[k] = torch.cat([p[k] for p in parts], dim=1)
            elif k != "ptr":
                out[k] = torch.cat([p[k] for p in parts], dim=0)
        for k, v in (y or {}).items():
            out[k] = torch.as_tensor(v).to(out["pos"].device)
        out["ptr"] = torch.from_numpy(ptr).to(out["pos"].device)
        return out
    # Four arrays cross PCIe (positions, cells, species, the two running sums of the crystals); one kernel derives what
    # the search and the model need per crystal and per atom (ops.graph_prep), two passes find the edges, one 16-byte
    # read-back sizes the outputs.  (The first device builder did this prologue with ~45 small library launches: 0.5 ms
    # of host time per call, and bounded the image loops per crystal instead of per pair.)
    n_atoms = int(ptr[-1])
    pair_ptr = np.zeros(n_crystals + 1, dtype=np.int64)
    np.cumsum(sizes * sizes, out=pair_ptr[1:])
    n_pairs = int(pair_ptr[-1])
    dev = torch.device(device)
    pos_d = torch.from_numpy(pos).to(dev)
    cell_d = torch.from_numpy(cell.reshape(-1, 9)).to(dev)
    ptrs_d = torch.from_numpy(np.stack([ptr, pair_ptr])).to(dev)
    ptr_d, pair_ptr_d = ptrs_d[0], ptrs_d[1]
    frac_d, bound_d, batch_d, pos32, cell32 = ops.graph_prep(pos_d, cell_d, ptr_d, r_cut)
    edge_index, shifts, num_neigh, pair_off, min_edges, csr = ops.neighbor_list(
        pos_d, cell_d, ptr_d, frac_d, bound_d, pair_ptr_d, r_cut, int(sizes.max()), n_pairs)
    if min_edges == 0:   # (came back with the edge count: no second sync on the common path)
        per_crystal = pair_off[pair_ptr_d[1:]] - pair_off[pair_ptr_d[:-1]]
        raise EdgelessStructures(torch.nonzero(per_crystal == 0).flatten().tolist())
    out = {
        "pos": pos32,
        "edge_index": edge_index,
        "edge_cell_shift": shifts,
        "cell": cell32,
        "num_neigh": num_neigh,
        "atomic_numbers": torch.from_numpy(Z).to(dev),
    }
    for k, v in (y or {}).items():
        out[k] = torch.as_tensor(v).to(dev)
    out["batch"] = batch_d
    out["ptr"] = ptr_d
    if csr is not None and EMIT_CSR:
        # the destination-sorted view every conv layer walks, emitted by the search itself (bit-identical to what
        # matten_csr_build derives from edge_index: tests/test_gpu_parity.py); the forward then skips that build.  Private
        # keys: whoever edits edge_index afterwards must drop them (nn/_nequip.ensure_graph trusts them when present)
        from ._key import AMD_PERM, AMD_ROWPTR, AMD_SRC

        out[AMD_PERM], out[AMD_ROWPTR], out[AMD_SRC] = csr
    return out
