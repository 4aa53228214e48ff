#!/usr/bin/env python3
"""Eager vs hipGraph-replayed inference forward at small batch sizes (fcc-64 crystals, paper model)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.graphs import GraphedForward
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

dev = "cuda:0"
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
torch.manual_seed(35)
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
for B in [int(b) for b in os.environ.get('BS', '1,8,32,64,125,250,1000').split(',')]:
    batch = collate(synthetic.fcc64_graphs(B), device=dev)
    with torch.no_grad():
        for _ in range(5): ref = model(dict(batch))[0]["elastic_tensor_full"]
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): model(dict(batch))
        torch.cuda.synchronize(); te = (time.perf_counter() - t) / 30
    g = GraphedForward(model, batch)
    out = g(batch); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30): g(batch)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 30
    print(f"B={B:5d}: eager {te*1e3:7.3f} ms   hipGraph replay {tg*1e3:7.3f} ms   max|diff| {(out - ref).abs().max().item():.2e}", flush=True)
