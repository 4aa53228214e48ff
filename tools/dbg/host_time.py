#!/usr/bin/env python3
"""Host-side enqueue time of one forward (input checks off: no sync) against its wall time, at 1 and 1000 crystals."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
dev = "cuda:0"
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
for B in (1, 1000):
    batch = collate(synthetic.fcc64_graphs(B), device=dev)
    with torch.no_grad():
        for _ in range(5): model(dict(batch))
        torch.cuda.synchronize()
        for m in model.modules():
            if hasattr(m, "check_species"): m.check_species = False   # no host sync: pure enqueue time
        t = time.perf_counter()
        for _ in range(20): model(dict(batch))
        t_enq = (time.perf_counter() - t) / 20
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t) / 20
        for m in model.modules():
            if hasattr(m, "check_species"): m.check_species = True
    print(f"B={B}: host enqueue {t_enq*1e3:.3f} ms per forward, wall {t_all*1e3:.3f} ms")
