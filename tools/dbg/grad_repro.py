import os, sys, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import LMAX2, build_pair
from test_gpu_training import _graphs
from matten_amd.data.graph import collate
DEV = "cuda:0"
graphs, ds = _graphs(os.path.join(ROOT, "tests", "golden"), 24)
_, model = build_pair(LMAX2, ds, randomize_bn=True)
model.train()
target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(5)).to(DEV)
runs = []
for _ in range(2):
    model.zero_grad(set_to_none=True)
    bn = {k: v.clone() for k, v in model.named_buffers()}
    preds, _ = model(collate(graphs, device=DEV))
    torch.nn.functional.mse_loss(preds["elastic_tensor_full"], target).backward()
    runs.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    with torch.no_grad():
        for k, v in model.named_buffers():
            v.copy_(bn[k])
for k in runs[0]:
    print("EQ " if torch.equal(runs[0][k], runs[1][k]) else "DIFF", k, f"{(runs[0][k]-runs[1][k]).abs().max().item():.2e}")
