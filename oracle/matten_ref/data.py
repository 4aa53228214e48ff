"""
CPU restatement of the reference's graph construction contract.  ORACLE / TEST INFRASTRUCTURE.

  neighbor_list   <- data/data.py:285-413 (``neighbor_list_and_relative_vec``), which calls
                     ase.neighborlist.primitive_neighbor_list("ijS", pbc=True, self_interaction=True)
                     (ase==3.23.0, un-vendored) and then drops edges with i==j and S==0.
                     Contract restated [ase-recalled]: every ordered triple (i, j, S) with
                     | r_j + S.cell - r_i | < r_cut  (strict), fp64, periodic in all 3 directions.
                     ASE's order within a centre atom is unspecified, so the oracle emits the
                     canonical order (i, j, Sx, Sy, Sz) and tests compare edge *sets*.
  crystal_graph   <- Crystal.from_points / DataPoint.__init__, data/data.py:44-144,217-279
  collate         <- torch_geometric Batch.from_data_list + DataPoint.tensor_property_to_dict
                     (data/data.py:146-159): the flat dict the backbone consumes
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

_SYMBOLS = (
    "X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
    "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
    "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr"
).split()
Z_OF = {s: z for z, s in enumerate(_SYMBOLS)}


def neighbor_list(pos: np.ndarray, cell: np.ndarray, r_cut: float):
    """Brute-force periodic neighbour list in fp64.  Returns (edge_index[2,E] i64, shifts[E,3] i64)."""
    pos = np.asarray(pos, dtype=np.float64)
    cell = np.asarray(cell, dtype=np.float64)
    n = len(pos)
    inv = np.linalg.inv(cell)
    frac = pos @ inv
    # perpendicular heights of the cell along each lattice direction
    vol = abs(np.linalg.det(cell))
    heights = np.array(
        [vol / np.linalg.norm(np.cross(cell[(a + 1) % 3], cell[(a + 2) % 3])) for a in range(3)]
    )
    spread = frac.max(axis=0) - frac.min(axis=0)
    nmax = np.ceil(r_cut / heights + spread + 1e-9).astype(int)
    rng = [np.arange(-m, m + 1) for m in nmax]
    S = np.stack(np.meshgrid(*rng, indexing="ij"), axis=-1).reshape(-1, 3)  # [ns,3]
    shift_vec = S @ cell  # [ns,3]
    # D[i,j,s] = pos[j] + shift[s] - pos[i]
    d = pos[None, :, None, :] + shift_vec[None, None, :, :] - pos[:, None, None, :]
    dist = np.sqrt((d * d).sum(-1))
    mask = dist < float(r_cut)
    zero = np.all(S == 0, axis=1)
    ii = np.arange(n)
    mask[ii, ii, np.nonzero(zero)[0][0]] = False
    i, j, s = np.nonzero(mask)
    shifts = S[s]
    order = np.lexsort((shifts[:, 2], shifts[:, 1], shifts[:, 0], j, i))
    i, j, shifts = i[order], j[order], shifts[order]
    if len(i) == 0:
        raise ValueError("After eliminating self edges, no edges remain in this system.")
    return np.stack([i, j]).astype(np.int64), shifts.astype(np.int64)


def crystal_graph(pos, cell, atomic_numbers, r_cut: float, y: Optional[Dict[str, torch.Tensor]] = None):
    """One crystal as the dict of tensors a ``Crystal`` data point holds."""
    pos = np.asarray(pos, dtype=np.float64)
    edge_index, shifts = neighbor_list(pos, cell, r_cut)
    n = len(pos)
    num_neigh = np.bincount(edge_index[0], minlength=n)
    d = {
        "pos": torch.as_tensor(pos, dtype=torch.float32),
        "edge_index": torch.as_tensor(edge_index, dtype=torch.int64),
        "edge_cell_shift": torch.as_tensor(shifts, dtype=torch.float32),
        "cell": torch.as_tensor(np.asarray(cell), dtype=torch.float32),
        "num_neigh": torch.as_tensor(num_neigh, dtype=torch.float32),
        "atomic_numbers": torch.as_tensor(np.asarray(atomic_numbers), dtype=torch.int64),
    }
    if y:
        d.update(y)
    return d


def collate(graphs: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Disjoint union of crystals -> backbone input dict."""
    out: Dict[str, List[torch.Tensor]] = {}
    batch, ptr = [], [0]
    off = 0
    for b, g in enumerate(graphs):
        n = g["pos"].shape[0]
        for k, v in g.items():
            if k == "edge_index":
                v = v + off
            out.setdefault(k, []).append(v)
        batch.append(torch.full((n,), b, dtype=torch.int64))
        off += n
        ptr.append(off)
    res = {}
    for k, vs in out.items():
        res[k] = torch.cat(vs, dim=1 if k == "edge_index" else 0)
    res["batch"] = torch.cat(batch)
    res["ptr"] = torch.tensor(ptr, dtype=torch.int64)
    return res


def structures_from_json(path: str):
    """Read a pandas-style JSON of pymatgen Structure dicts (the reference's dataset format,
    dataset/structure_scalar_tensor.py:229-243) without pymatgen.  Returns a list of dicts
    {lattice[3,3], cart_coords[N,3], atomic_numbers[N], (elastic_tensor_full)}."""
    with open(path) as f:
        raw = json.load(f)
    keys = sorted(raw["structure"].keys(), key=lambda s: int(s))
    out = []
    for k in keys:
        s = raw["structure"][k]
        lat = np.array(s["lattice"]["matrix"], dtype=np.float64)
        xyz = np.array([site["xyz"] for site in s["sites"]], dtype=np.float64)
        zs = []
        for site in s["sites"]:
            assert len(site["species"]) == 1, "disordered site"
            zs.append(Z_OF[site["species"][0]["element"]])
        d = {"lattice": lat, "cart_coords": xyz, "atomic_numbers": np.array(zs, dtype=np.int64)}
        if "elastic_tensor_full" in raw:
            d["elastic_tensor_full"] = np.array(raw["elastic_tensor_full"][k], dtype=np.float64)
        out.append(d)
    return out
