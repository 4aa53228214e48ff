import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER, build_pair
from matten_amd.data import synthetic
from matten_amd.data.graph import collate, crystal_graph
DEV = "cuda:0"
lone = crystal_graph(np.array([[0.0, 0, 0], [1.5, 0, 0], [6.0, 6.0, 6.0]]), 12.0 * np.eye(3), [29, 79, 29], 5.0)
which = sys.argv[1] if len(sys.argv) > 1 else "lone"
graphs = synthetic.fcc64_graphs(2) + ([lone] if which == "lone" else [])
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
hp = dict(PAPER)
ref, fused = build_pair(hp, ds, randomize_bn=True)
os.environ["MATTEN_CONV_FUSED"] = "0"
_, plain = build_pair(hp, ds, randomize_bn=True)
del os.environ["MATTEN_CONV_FUSED"]
plain.load_state_dict(fused.state_dict())
cpu, a, b = collate(graphs), collate(graphs, device=DEV), collate(graphs, device=DEV)
with torch.no_grad():
    for (name, rmod), (_, fmod), (_, pmod) in zip(ref.backbone.named_children(), fused.backbone.named_children(), plain.backbone.named_children()):
        cpu, a, b = rmod(cpu), fmod(a), pmod(b)
        if "node_features" in cpu:
            fa, fb, fc = a["node_features"].cpu(), b["node_features"].cpu(), cpu["node_features"]
            print(name, "fused nan", int(torch.isnan(fa).sum()), "plain nan", int(torch.isnan(fb).sum()), "max", float(fc.abs().max()),
                  "f-p", float((fa - fb).abs().nan_to_num(0).max()), "p-ref", float((fb - fc).abs().nan_to_num(0).max()),
                  "f-ref", float((fa - fc).abs().nan_to_num(0).max()))
            if torch.isnan(fa).any():
                rows = torch.isnan(fa).any(1).nonzero().flatten()
                print("   fused NaN rows", rows[:20].tolist(), "cols", torch.isnan(fa).any(0).nonzero().flatten()[:20].tolist())
            if torch.isnan(fb).any():
                rows = torch.isnan(fb).any(1).nonzero().flatten()
                print("   plain NaN rows", rows[:20].tolist())
