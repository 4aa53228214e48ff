import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER, build_pair
from test_gpu_training import _graphs
from matten_amd.data.graph import collate
DEV = "cuda:0"
graphs, ds = _graphs(os.path.join(ROOT, "tests", "golden"), 6)
ref, model = build_pair(PAPER, ds, randomize_bn=True)
ref.train(); model.train()
target = torch.randn(6, 21, generator=torch.Generator().manual_seed(7))
torch.nn.functional.mse_loss(ref.decode(collate(graphs)), target).backward()
torch.nn.functional.mse_loss(model(collate(graphs, device=DEV))[0]["elastic_tensor_full"], target.to(DEV)).backward()
named = dict(model.named_parameters())
for k, p in ref.named_parameters():
    if p.grad is None: continue
    g, gm = p.grad.double(), named[k].grad.cpu().double()
    mx = g.abs().max().item()
    err = (gm - g).abs()
    i = err.argmax()
    flips = ((gm * g) < 0) & (g.abs() > 1e-4 * mx)
    if "weight_nn.layer2" in k or flips.any():
        print(f"{k:60s} max|g| {mx:.3e} max err {err.max().item():.3e} ({err.max().item()/mx:.1e} rel) sign flips among solid {int(flips.sum())}"
              + (f" e.g. ref {g.flatten()[flips.flatten().nonzero()[0]].item():.3e} ours {gm.flatten()[flips.flatten().nonzero()[0]].item():.3e}" if flips.any() else ""))
