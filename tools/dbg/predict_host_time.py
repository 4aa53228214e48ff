#!/usr/bin/env python3
"""predict() end to end by stage (host packing / species check / device graphs + forward / results)"""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
structs = synthetic.fcc64_structures(1000)
cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
P.predict(structs[:16], model=model, config=cfg)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); pos, cell, Z, ptr, keep, failed = P.pack_structures(structs); t1 = T()
    P.check_species(model, structs, Z=Z, ptr=ptr, index=keep); t2 = T()
    out, edgeless = P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200); t3 = T()
    res = [out[i] for i in range(len(out))]; t4 = T()
    t5 = T(); P.predict(structs, model=model, config=cfg, batch_size=200); t6 = T()
    print(f"pack {1e3*(t1-t0):.2f}  species {1e3*(t2-t1):.2f}  evaluate_soa {1e3*(t3-t2):.2f}  list {1e3*(t4-t3):.2f}  | predict() {1e3*(t6-t5):.2f} ms")
for budget in (65536, 32768, 16384, 8192):
    P.NODE_BUDGET = budget
    P.predict(structs, model=model, config=cfg, batch_size=200)
    t0 = T()
    for _ in range(3): P.predict(structs, model=model, config=cfg, batch_size=200)
    print(f"node budget {budget}: predict() {1e3*(T()-t0)/3:.2f} ms per 1000 fcc-64")
