#!/usr/bin/env python3
"""batch_graphs_gpu_soa on 1000 fcc-64 crystals (and the n100 sample x 10): wall time per call, synchronised"""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.data.graph import batch_graphs_gpu_soa
from matten_amd.data.io import structures_from_json

def T():
    torch.cuda.synchronize(); return time.perf_counter()

sets = {"fcc64 x 1000": synthetic.fcc64_structures(1000),
        "n100 x 10": structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json")) * 10}
for name, structs in sets.items():
    pos, cell, Z, ptr, keep, failed = P.pack_structures(structs)
    for _ in range(3):
        g = batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda:0")
    n = 20
    t0 = T()
    for _ in range(n):
        g = batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda:0")
    dt = (T() - t0) / n
    print(f"{name}: {1e3*dt:.3f} ms per build ({g['pos'].shape[0]} atoms, {g['edge_index'].shape[1]} edges)")
