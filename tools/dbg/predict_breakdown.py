#!/usr/bin/env python3
"""Where evaluate_soa's time goes at 1000 fcc-64 structures: device graph build, forward, to_cartesian, D2H (each synchronised)."""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.data.graph import batch_graphs_gpu_soa
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.utils import CartesianTensorWrapper
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
structs = synthetic.fcc64_structures(1000)
pos, cell, Z, ptr, keep, failed = P.pack_structures(structs)
conv = CartesianTensorWrapper("ijkl=jikl=klij")
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); batch = batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda:0"); t1 = T()
    with torch.no_grad():
        p = model(batch, task_name="elastic_tensor_full")[0]["elastic_tensor_full"]; t2 = T()
        c = conv.to_cartesian(p); t3 = T()
        out = torch.full((1000, 3, 3, 3, 3), float("nan"), device="cuda:0"); out[torch.arange(1000, device="cuda:0")] = c; t4 = T()
        h = out.cpu().numpy(); t5 = T()
    print(f"graphs {1e3*(t1-t0):.2f}  forward {1e3*(t2-t1):.2f}  to_cartesian {1e3*(t3-t2):.2f}  scatter {1e3*(t4-t3):.2f}  d2h {1e3*(t5-t4):.2f} ms")
