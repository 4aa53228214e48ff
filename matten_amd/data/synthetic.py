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


def fcc64_structures(n: int, seed: int = FCC_SEED, start: int = 0) -> List[Dict[str, np.ndarray]]:
    """crystals [start, start + n) of the set seeded with `seed` (one sequential random stream: a shard of a larger set
    is the same crystals the whole set would hold at those indices -- SURVEY.md 8d config 5: rank r of an 8-GPU run owns
    crystals [1000 r, 1000 r + 1000) of ONE 8000-crystal set)"""
    rng = np.random.default_rng(seed)
    prim = 0.5 * np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 1.0], [1.0, 1.0, 0.0]])
    grid = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    out = []
    for _ in range(start + n):
        a = rng.uniform(4.20, 4.90)
        cell = 4.0 * a * prim
        pos = (grid @ (a * prim)) + rng.normal(0.0, 0.02, size=(64, 3))
        z = rng.choice(FCC_METALS, size=64)
        out.append({"lattice": cell, "cart_coords": pos, "atomic_numbers": z.astype(np.int64)})
    return out[start:]


def fcc64_graphs(n: int, seed: int = FCC_SEED, r_cut: float = 5.0, start: int = 0):
    return [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], r_cut)
            for s in fcc64_structures(n, seed, start)]


def fcc64_shard(rank: int, world: int, per_rank: int, r_cut: float = 5.0):
    """The crystals rank `rank` of `world` owns (bench.py, BASELINE configs[2] / configs[4]): one GPU runs the config-3
    set (seed FCC_SEED); N > 1 GPUs shard ONE set of N x per_rank crystals seeded FCC_SEED + 1 contiguously by batch
    index, rank r taking [r per_rank, (r + 1) per_rank)."""
    if world == 1:
        return fcc64_graphs(per_rank, FCC_SEED, r_cut)
    return fcc64_graphs(per_rank, FCC_SEED + 1, r_cut, start=rank * per_rank)


def tile_batch(unique: List[Dict[str, torch.Tensor]], n_total: int) -> List[Dict[str, torch.Tensor]]:
    """Repeat a pool of distinct crystals up to n_total graphs (graph construction is host work outside the timed path)."""
    return [unique[i % len(unique)] for i in range(n_total)]
