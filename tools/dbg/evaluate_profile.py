#!/usr/bin/env python3
"""evaluate_soa on 1000 fcc-64 crystals: stage times (synchronised) and the host profile of one call"""
import cProfile, os, pstats, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.data.graph import batch_graphs_gpu_soa
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
structs = synthetic.fcc64_structures(1000)
pos, cell, Z, ptr, keep, failed = P.pack_structures(structs)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for _ in range(3): P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200)
n = 10
t0 = T()
for _ in range(n): g = batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda:0")
t1 = T()
with torch.no_grad():
    for _ in range(n): p = model(dict(g), task_name="elastic_tensor_full")[0]["elastic_tensor_full"]
t2 = T()
for _ in range(n): out = P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200)
t3 = T()
print(f"build {1e3*(t1-t0)/n:.3f}  forward {1e3*(t2-t1)/n:.3f}  evaluate_soa {1e3*(t3-t2)/n:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
