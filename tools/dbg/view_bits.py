import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER, build_pair
from matten_amd.data import synthetic
from matten_amd.data.graph import batch_graphs_gpu
from matten_amd.nn import conv as pconv
DEV = "cuda:0"
n = 200
structs = synthetic.fcc64_structures(n)
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
ref, model = build_pair(PAPER, ds, randomize_bn=True)
triples = [(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in structs]
pick = [0, 1, 137]
def feats(tr, km_rows):
    pconv.AGG_KM_MIN_ROWS = km_rows
    data = batch_graphs_gpu(tr, 5.0, DEV)
    out = {}
    with torch.no_grad():
        for name, mod in model.backbone.named_children():
            data = mod(data)
            out[name] = data["node_features"].clone() if "node_features" in data else None
    return out
big = feats(triples, 8192)
small = feats([triples[i] for i in pick], 8192)
big_rows = feats(triples, 10**9)
rows = torch.cat([torch.arange(64 * i, 64 * i + 64) for i in pick]).to(DEV)
for name in big:
    if big[name] is None: continue
    a, b, c = big[name][rows], small[name], big_rows[name][rows]
    print(f"{name:24s} km-vs-small equal {torch.equal(a, b)} maxdiff {(a-b).abs().max().item():.3e} | rowpath-vs-small equal {torch.equal(c, b)}")
v = model.backbone._modules["conv_layer_last"]._view
print("view agg_plan", v.agg_plan is not None, "d_mid", v.tp.plan.d_mid, "lin2 w_stride", v.lin2.plan.w_stride, "passes", len(v.lin2.plan.passes))
