#!/usr/bin/env python3
"""End-to-end matten_amd.predict.predict() on 1000 fcc-64 structures (dict inputs), with a breakdown."""
import os, sys, time, tempfile
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
torch.manual_seed(0)
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
structs = synthetic.fcc64_structures(n)
P.predict(structs[:8], model=model, config=cfg)  # warm-up
for bs in (200, 1000):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = P.predict(structs, model=model, config=cfg, batch_size=bs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"predict({n} structures, batch_size={bs}): {dt*1e3:.1f} ms -> {n/dt:.0f} crystals/s end to end")
t = time.perf_counter(); pos, cell, Z, ptr, keep, failed = P.pack_structures(structs); t1 = time.perf_counter()
P.check_species(model, structs, Z, ptr, keep); t2 = time.perf_counter()
tens, _ = P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=1000); t3 = time.perf_counter()
print(f"  pack_structures {1e3*(t1-t):.1f} ms, check_species {1e3*(t2-t1):.1f} ms, evaluate_soa (graphs + forward + D2H) {1e3*(t3-t2):.1f} ms")
