#!/usr/bin/env python3
"""cProfile of the EAGER n100 forward (configs[1]: 100 crystals, 473 atoms): where the host spends its time between launches.
   python3 tools/dbg/n100_host_profile.py [n_forwards]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data.graph import batch_graphs_gpu
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

dev = "cuda:0"
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
torch.manual_seed(35)
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams={"allowed_species": species, "average_num_neighbors": 30.4017}).to(dev).eval()
batch = batch_graphs_gpu([(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in n100], 5.0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with torch.no_grad():
    for _ in range(20):
        model(dict(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(dict(batch))
    torch.cuda.synchronize()
    print(f"eager: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward")
    # host enqueue time alone: no synchronisation inside the loop, time until the last launch is queued
    t0 = time.perf_counter()
    for _ in range(n):
        model(dict(batch))
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print(f"host enqueue: {t_host * 1e3:.3f} ms per forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        model(dict(batch))
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.print_callers("parameters")
st.print_callers("named_modules")
st.print_callers("tolist")
