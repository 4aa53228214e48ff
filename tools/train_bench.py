#!/usr/bin/env python3
"""Times one optimisation step (fwd + bwd + Adam) of BASELINE.json configs[3]:
lmax=2, 3 gated blocks, batch 32, BatchNorm train mode, MSE in irreps space, on a synthetic set with the
size distribution of the reference's n100 example (the Zenodo set is not in the reference tree)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import LMAX2
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.data.io import structures_from_json

structs = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in structs]
species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
ds = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
torch.manual_seed(3)
model = ScalarTensorModel(backbone_hparams=dict(LMAX2), dataset_hparams=ds).to("cuda:0").train()
FUSED = os.environ.get("ADAM", "fused") == "fused"   # one multi-tensor kernel for all 60-odd parameters
opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-5, fused=FUSED)
BS = int(os.environ.get("BATCH", "32"))
pool = [graphs[i % len(graphs)] for i in range(3 * BS)]
batches = [collate(pool[i:i + BS], device="cuda:0") for i in range(0, 3 * BS, BS)]
targets = [torch.randn(BS, 21, device="cuda:0") for _ in batches]
def step(b, t):
    preds, _ = model(dict(b))
    loss = torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(3):
    for b, t in zip(batches, targets): step(b, t)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for _ in range(10):
    for b, t in zip(batches, targets): loss = step(b, t); n += 1
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
nodes = sum(int(b["pos"].shape[0]) for b in batches) / len(batches); edges = sum(int(b["edge_index"].shape[1]) for b in batches) / len(batches)
print(f"training step (batch {BS} crystals, {nodes:.0f} nodes, {edges:.0f} edges avg, Adam {'fused' if FUSED else 'per-tensor'}): "
      f"{dt*1e3:.2f} ms/step, {BS/dt:.0f} crystals/s, final loss {loss.item():.4f}")
if os.environ.get("PROFILE"):   # host profile of the eager step (the step is launch-bound at this size)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        for b, t in zip(batches, targets): step(b, t)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
