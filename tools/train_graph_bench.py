#!/usr/bin/env python3
"""One optimisation step of configs[3] (batch 32, launch-bound: ~350 launches for 2.8 ms of kernel time) captured in a
hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm): eager vs replay, same batch shapes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import LMAX2
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.data.io import structures_from_json
from matten_amd.graphs import GraphedTrainStep
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

structs = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in structs]
species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
ds = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
BS = int(os.environ.get("BATCH", "32"))
batch = collate(graphs[:BS], device="cuda:0")
target = torch.randn(BS, 21, device="cuda:0")

def make():
    torch.manual_seed(3)
    m = ScalarTensorModel(backbone_hparams=dict(LMAX2), dataset_hparams=ds).to("cuda:0").train()
    o = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5, fused=True, capturable=True)
    return m, o

def loss_fn(preds, t):
    return torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)

# eager
model, opt = make()
def eager():
    preds, _ = model(dict(batch))
    loss = loss_fn(preds, target)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(5): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): le = eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30

# graph
model2, opt2 = make()
gs = GraphedTrainStep(model2, opt2, loss_fn, batch, target, warmup=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): lg = gs.step(batch, target)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 30
print(f"batch {BS}: eager {te*1e3:.2f} ms/step (loss {le.item():.5f})   hipGraph replay {tg*1e3:.2f} ms/step (loss {lg.item():.5f})")
# same trajectory: 10 steps each from the same seed (the graph's warm-up steps are rolled back by GraphedTrainStep); with
# no atomics anywhere on the step the two runs agree to the last bit
m_a, o_a = make()
m_b, o_b = make()
g_b = GraphedTrainStep(m_b, o_b, loss_fn, batch, target, warmup=3)
for _ in range(10):
    preds, _ = m_a(dict(batch))
    l = loss_fn(preds, target)
    o_a.zero_grad(); l.backward(); o_a.step()
    g_b.step(batch, target)
d = max((a - b).abs().max().item() for a, b in zip(m_a.parameters(), m_b.parameters()))
print("max |param eager - param graph| after 10 steps each:", d)
