"""n100 forward as a hipGraph replay: ms per forward (environment knobs A/B: MATTEN_AGG_KM_MIN_ROWS, MATTEN_HUB_SPLIT_LEN, ...)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data.graph import batch_graphs_gpu
from matten_amd.data.io import structures_from_json
from matten_amd.graphs import GraphedForward
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
torch.manual_seed(35)
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams={"allowed_species": species, "average_num_neighbors": 30.4}).to("cuda:0").eval()
batch = batch_graphs_gpu([(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in n100], 5.0, "cuda:0")
with torch.no_grad():
    g = GraphedForward(model, batch)
    for _ in range(10):
        g(batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        g(batch)
    torch.cuda.synchronize()
    print(f"n100 hipGraph forward: {(time.perf_counter() - t) / 200 * 1e3:.4f} ms")
