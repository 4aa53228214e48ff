import os, sys, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import LMAX2
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
structs = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in structs[:32]]
species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
ds = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
model = ScalarTensorModel(backbone_hparams=dict(LMAX2), dataset_hparams=ds).to("cuda:0").train()
opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-5, fused=True)
b = collate(graphs, device="cuda:0"); t = torch.randn(32, 21, device="cuda:0")
def step():
    preds, _ = model(dict(b)); loss = torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
import traceback
calls = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if "matten_amd" in x.filename]
        calls[(name, (fr[-1].filename.split("matten_amd/")[-1] + ":" + str(fr[-1].lineno)) if fr else "?")] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for n in ("zeros", "zeros_like", "ones", "full", "empty_like"):
    wrap(torch, n)
step()
for (name, where), n in sorted(calls.items(), key=lambda kv: -kv[1]):
    if name != "empty_like": print(f"py {n:4d} {name:12s} {where}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::ones", "aten::mul", "aten::add", "aten::add_", "aten::mul_", "aten::copy_", "aten::index", "aten::sum"):
        st = [f for f in (ev.stack or []) if "matten_amd" in f or "tools/" in f or "autograd" in f.lower()]
        cnt[(ev.name, st[0][-90:] if st else "(no python frame: autograd engine)")] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:4d} {name:14s} {where}")
