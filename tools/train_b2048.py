#!/usr/bin/env python3
"""One optimisation step of the lmax-2 configuration at batch 2048 (the n100 sample tiled): step time and the HIP-event
times of its big kernels.   python3 tools/train_b2048.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd import ops
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = "cuda:0"
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
lmax2 = dict(PAPER_HPARAMS, irreps_edge_sh="0e + 1o + 2e", conv_layer_irreps="32x0o+32x0e+16x1o+16x1e+4x2o+4x2e")
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in n100]
ds4 = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
BL = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tbl = collate([graphs[i % len(graphs)] for i in range(BL)], device=dev)
target = torch.randn(BL, 21, device=dev)
torch.manual_seed(3)
m = ScalarTensorModel(backbone_hparams=dict(lmax2), dataset_hparams=ds4).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5, fused=True, capturable=True)


def step():
    loss = torch.nn.functional.mse_loss(m(dict(tbl))[0]["elastic_tensor_full"], target)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ops.enable_event_timing(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
ev = {k: sum(v) / len(v) for k, v in ops.event_timings_ms().items()}
ops.enable_event_timing(False)
print(f"batch {BL}: {dt*1e3:.3f} ms per step, loss {float(l):.4f}")
groups = {}
for k, v in ev.items():
    groups.setdefault(k.split("/")[0], []).append((k, v))
for g, items in sorted(groups.items(), key=lambda kv: -sum(v for _, v in kv[1])):
    print(f"  {g:28s} {sum(v for _, v in items):7.3f} ms  " + "  ".join(f"{k.split('/',1)[1] if '/' in k else ''}:{v:.3f}" for k, v in items))
