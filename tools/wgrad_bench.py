#!/usr/bin/env python3
"""Weight-gradient kernel of the species linear (matten_species_linear_wgrad) on the lmax-2 training configuration at
batch 2048 (the n100 sample tiled: 9652 rows, 73 species): microseconds per call for every species linear of the model.
    python3 tools/wgrad_bench.py [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd import ops
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.nn.utils import SpeciesLinear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = "cuda:0"
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
lmax2 = dict(PAPER_HPARAMS, irreps_edge_sh="0e + 1o + 2e", conv_layer_irreps="32x0o+32x0e+16x1o+16x1e+4x2o+4x2e")
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in n100]
ds4 = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
BL = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tbl = collate([graphs[i % len(graphs)] for i in range(BL)], device=dev)
m = ScalarTensorModel(backbone_hparams=dict(lmax2), dataset_hparams=ds4).to(dev).train()
lut = torch.full((200,), -1, dtype=torch.long, device=dev)
lut[torch.tensor(species, device=dev)] = torch.arange(len(species), device=dev)
sidx = lut[tbl["atomic_numbers"].long()]
N = sidx.shape[0]
order = torch.argsort(sidx, stable=True).int()
seg = torch.zeros(len(species) + 1, dtype=torch.int32, device=dev)
seg[1:] = torch.cumsum(torch.bincount(sidx, minlength=len(species)), 0).int()
print(f"{N} rows, {len(species)} species, largest species {int((seg[1:] - seg[:-1]).max())} rows, "
      f"scratch floats per unit of w_stride {ops._lib.load().matten_species_linear_wgrad_scratch_floats(N, len(species), 1)}")
total = 0.0
for name, mod in m.named_modules():
    if not isinstance(mod, SpeciesLinear):
        continue
    p = mod.plan
    x = torch.randn(N, p.d_in, device=dev)
    dy = torch.randn(N, p.d_out, device=dev)
    segs = [mod._tables.get(f"meta{i}", dev) for i in range(len(p.passes))]
    so = (order, seg)
    for _ in range(3):
        ops.species_linear_wgrad(x, dy, so, len(species), segs, p.w_stride)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.species_linear_wgrad(x, dy, so, len(species), segs, p.w_stride)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    total += us
    print(f"  {name:44s} d_in {p.d_in:5d} d_out {p.d_out:4d} segs {sum(len(s) for s in p.passes):2d}  {us:8.1f} us")
print(f"  sum {total:.1f} us")
