import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.graphs import GraphedForward
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to("cuda:0").eval()
graphs = synthetic.fcc64_graphs(120)
out = []
for B in (4, 11, 30, 60, 120):
    batch = collate(graphs[:B], device="cuda:0")
    with torch.no_grad():
        g = GraphedForward(model, batch)
        for _ in range(10): g(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): g(batch)
        torch.cuda.synchronize()
    out.append(f"B={B} ({64*B} atoms): {(time.perf_counter() - t) / 50 * 1e3:.3f} ms")
print(f"HUB_SPLIT_LEN={os.environ.get('MATTEN_HUB_SPLIT_LEN', 'default')}: " + "  ".join(out))
