#!/usr/bin/env python3
"""Random small models (irreps, multiplicities, lmax, depth, normalisation, neighbour normalisation) on random small
crystals: product on the GPU against the oracle on the CPU.  Test infrastructure (it drives the oracle): tests/fuzz_models.py [n_cases] [seed]"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER, build_pair
from matten_amd.data.graph import collate, crystal_graph

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    lmax = int(rng.integers(1, 5))
    irr = []
    for l in range(lmax + 1):
        for p in "oe":
            if rng.random() < 0.75 or (l == 0 and p == "e"):
                irr.append(f"{int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 32]))}x{l}{p}")
    hp = dict(PAPER)
    hp["irreps_edge_sh"] = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lmax + 1))
    hp["conv_layer_irreps"] = "+".join(irr)
    hp["num_layers"] = int(rng.integers(1, 4))
    hp["species_embedding_dim"] = int(rng.choice([4, 16, 33]))
    hp["num_radial_basis"] = int(rng.choice([4, 8, 10]))
    hp["normalization"] = rng.choice(["batch", None])
    hp["average_num_neighbors"] = rng.choice(["auto", None])
    # crystals: random triclinic cells with 1-6 atoms
    graphs, species = [], sorted(rng.choice(np.arange(1, 90), size=int(rng.integers(1, 5)), replace=False).tolist())
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 7))
        cell = 3.0 * np.eye(3) + rng.normal(0, 0.6, (3, 3))
        pos = rng.random((n, 3)) @ cell
        graphs.append(crystal_graph(pos, cell, rng.choice(species, size=n), 5.0))
    ds = {"allowed_species": species, "average_num_neighbors": float(np.mean([g["num_neigh"].mean() for g in graphs]))}
    try:
        ref, model = build_pair(hp, ds, randomize_bn=True, seed=case)
    except (ValueError, RuntimeError, NotImplementedError) as e:  # e.g. no path to the gates: both sides must refuse
        print(case, "construction refused:", type(e).__name__, str(e)[:80])
        continue
    with torch.no_grad():
        want = ref.decode(collate(graphs))
        got = model.decode(collate(graphs, device="cuda:0"))["elastic_tensor_full"].cpu()
    scale = max(1e-6, want.abs().max().item())
    err = (got - want).abs().max().item() / scale
    flag = "" if err < 5e-4 else "  <-- BAD"
    bad += bool(flag)
    print(f"{case:3d} lmax {lmax} layers {hp['num_layers']} irreps {hp['conv_layer_irreps'][:60]:60s} rel err {err:.1e}{flag}")
print("bad", bad)
sys.exit(1 if bad else 0)
