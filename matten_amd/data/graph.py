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

import numpy as np
import torch
from scipy.spatial import cKDTree


def neighbor_list(pos: np.ndarray, cell: np.ndarray, r_cut: float):
    """-> edge_index [2,E] int64, shifts [E,3] int64 (canonical order)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    n = pos.shape[0]
    rc = float(r_cut)
    # number of periodic images needed along each lattice direction: distance between lattice planes
    recip = np.linalg.inv(cell).T  # rows: reciprocal vectors (without 2 pi)
    plane_dist = 1.0 / np.linalg.norm(recip, axis=1)
    frac = pos @ np.linalg.inv(cell)
    span = frac.max(0) - frac.min(0) if n else np.zeros(3)
    reach = np.ceil(rc / plane_dist + span).astype(int)

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
