#!/usr/bin/env python3
"""
Regenerates tests/golden/oracle_golden.npz.

The reference's own tests store no numeric output of the conv stack and e3nn cannot be imported in
the build container, so these vectors are produced by the CPU oracle itself
and pin it against regressions; they are NOT e3nn outputs ("parity unpinned").  Inputs are the
data files the reference's tests/datasets hold (copied verbatim next to this script):
  elastic_tensor_one.json                           reference tests/test_files/
  example_crystal_elasticity_tensor_n100.json       reference datasets/
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import EQUIV_TEST, PAPER  # noqa: E402
from oracle.e3nn_lite import io, nn as enn, o3  # noqa: E402
from oracle.matten_ref import data as rdata  # noqa: E402
from oracle.matten_ref.model import ScalarTensorOracle  # noqa: E402


def main():
    torch.set_num_threads(4)
    out = {}
    # constants
    out["normalize2mom"] = np.array(
        [enn.normalize2mom(f).cst for f in (torch.nn.functional.silu, torch.sigmoid, torch.tanh, torch.abs)]
    )
    out["w3j_111"] = o3.wigner_3j(1, 1, 1, dtype=torch.float64).numpy()
    out["w3j_224"] = o3.wigner_3j(2, 2, 4, dtype=torch.float64).numpy()
    out["w3j_444"] = o3.wigner_3j(4, 4, 4, dtype=torch.float64).numpy()
    out["cart_basis_ijkl"] = io.CartesianTensor("ijkl=jikl=klij").change_of_basis(torch.float64).numpy()
    v = torch.tensor([[0.3, -0.5, 0.8124038404635961], [1.0, 2.0, -0.5], [0.0, 0.0, 2.0]], dtype=torch.float64)
    out["sh_points"] = v.numpy()
    out["sh_values"] = o3.spherical_harmonics([0, 1, 2, 3, 4], v, True, "component").numpy()

    # TeO fixture, reference test hparams, seed 35
    s = rdata.structures_from_json(os.path.join(HERE, "elastic_tensor_one.json"))[0]
    g = rdata.crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0)
    torch.manual_seed(35)
    m = ScalarTensorOracle(dict(EQUIV_TEST), {"allowed_species": [8, 52]}).eval()
    with torch.no_grad():
        out["teo_cartesian"] = m.decode(rdata.collate([g])).numpy()
    out["teo_edge_index"] = g["edge_index"].numpy()

    # n100, paper hparams, first 6 crystals, seed 35
    structs = rdata.structures_from_json(os.path.join(HERE, "example_crystal_elasticity_tensor_n100.json"))
    graphs = [rdata.crystal_graph(t["cart_coords"], t["lattice"], t["atomic_numbers"], 5.0) for t in structs]
    species = sorted({int(z) for t in structs for z in t["atomic_numbers"]})
    avg = float(torch.cat([gr["num_neigh"] for gr in graphs]).mean())
    out["n100_species"] = np.array(species)
    out["n100_edges_per_crystal"] = np.array([gr["edge_index"].shape[1] for gr in graphs])
    out["n100_avg_num_neigh"] = np.array(avg)
    torch.manual_seed(35)
    m = ScalarTensorOracle(dict(PAPER), {"allowed_species": species, "average_num_neighbors": avg}).eval()
    with torch.no_grad():
        out["n100_first6_irreps"] = m.decode(rdata.collate(graphs[:6])).numpy()
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
