#!/usr/bin/env python3
"""predict() on 4000 fcc-64 structures (and 4000 n100-sized ones): one slab against slabs of 1024 packed behind the device"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
sets = {"fcc-64": (synthetic.fcc64_structures(4000), list(synthetic.FCC_METALS), 18.0),
        "n100-sized": (n100 * 40, sorted({int(z) for s in n100 for z in s["atomic_numbers"]}), 30.4)}
for name, (structs, species, avg) in sets.items():
    model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams={"allowed_species": species, "average_num_neighbors": avg}).to("cuda:0").eval()
    for slab in (10**9, 1024):
        P.PREDICT_SLAB = slab
        P.predict(structs[:1500], model=model, config=cfg, batch_size=200)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = P.predict(structs, model=model, config=cfg, batch_size=200)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{name}: slab {slab if slab < 10**9 else 'all'}: {1e3*dt/4:.2f} ms per 1000 structures, {4000/dt:.0f} crystals/s")
