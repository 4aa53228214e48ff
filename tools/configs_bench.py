#!/usr/bin/env python3
"""BASELINE.json configs[0] (Si diamond) and configs[1] (the reference's n=100 example set): forward latency and
predict() end to end, random-init paper model (the checkpoint is not shipped)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data.graph import batch_graphs_gpu
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
a = 5.46
si = {"lattice": np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]]),
      "cart_coords": np.array([[0.0, 0, 0], [a / 4, a / 4, a / 4]]), "atomic_numbers": np.array([14, 14])}
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
for name, structs in (("configs[0] Si diamond (2 atoms, 56 edges)", [si]), ("configs[1] n100 (473 atoms, 14380 edges)", n100)):
    species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
    torch.manual_seed(35)
    model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams={"allowed_species": species, "average_num_neighbors": 30.4}).to("cuda:0").eval()
    batch = batch_graphs_gpu([(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in structs], 5.0, "cuda:0")
    with torch.no_grad():
        for _ in range(5): model.decode(dict(batch))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): model.decode(dict(batch))
        torch.cuda.synchronize(); fwd = (time.perf_counter() - t) / 50
    P.predict(structs, model=model, config=cfg)
    t = time.perf_counter()
    for _ in range(10): P.predict(structs, model=model, config=cfg)
    e2e = (time.perf_counter() - t) / 10
    print(f"{name}: forward {fwd*1e3:.2f} ms ({len(structs)/fwd:.0f} crystals/s), predict() end to end {e2e*1e3:.2f} ms")
