"""
Synthetic workloads named by BASELINE.json / SURVEY.md section 8(d).

fcc-64: 4x4x4 primitive fcc cells (64 atoms), a ~ U[4.20, 4.90] A so that exactly the first two
neighbour shells fall inside the 5 A cutoff (18 neighbours/atom, 1152 edges), Gaussian jitter
sigma = 0.02 A, species i.i.d. from ten fcc metals.  Seeded with numpy default_rng(20250711).
"""
from typing import Dict, List

import numpy as np
import torch

from .graph import collate, crystal_graph

FCC_METALS = (13, 28, 29, 45, 46, 47, 77, 78, 79, 82)
FCC_SEED = 20250711


def fcc64_structures(n: int, seed: int = FCC_SEED) -> List[Dict[str, np.ndarray]]:
    rng = np.random.default_rng(seed)
    prim = 0.5 * np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 1.0], [1.0, 1.0, 0.0]])
    grid = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    out = []
    for _ in range(n):
        a = rng.uniform(4.20, 4.90)
        cell = 4.0 * a * prim
        pos = (grid @ (a * prim)) + rng.normal(0.0, 0.02, size=(64, 3))
        z = rng.choice(FCC_METALS, size=64)
        out.append({"lattice": cell, "cart_coords": pos, "atomic_numbers": z.astype(np.int64)})
    return out


def fcc64_graphs(n: int, seed: int = FCC_SEED, r_cut: float = 5.0):
    return [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], r_cut) for s in fcc64_structures(n, seed)]


def tile_batch(unique: List[Dict[str, torch.Tensor]], n_total: int) -> List[Dict[str, torch.Tensor]]:
    """Repeat a pool of distinct crystals up to n_total graphs (graph construction is host work outside the timed path)."""
    return [unique[i % len(unique)] for i in range(n_total)]
