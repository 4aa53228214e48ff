#!/usr/bin/env python3
"""tp_fused on the last conv layer (fcc-64 x B crystals), one launch per group kind: where the kernel's time goes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matten_amd import ops, plan as mplan
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.o3 import Irreps

B = int(os.environ.get("B", 1000))
dev = "cuda:0"
graphs = synthetic.fcc64_graphs(min(B, 64))
graphs = [graphs[i % len(graphs)] for i in range(B)]
b = collate(graphs, device=dev)
N, E = b["pos"].shape[0], b["edge_index"].shape[1]
irr = "32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e"
# TARGET=view: the last conv layer as inference runs it (dead-output elimination: the head's irreps only)
target = "32x0e+4x2e+2x4e" if os.environ.get("TARGET") == "view" else irr
p = mplan.plan_uvu(irr, Irreps.spherical_harmonics(4), target)
perm, rowptr, src, _ = ops.csr_build(b["edge_index"], N)
geo = ops.edge_geom(b["pos"], b["edge_index"], b["edge_cell_shift"], b["cell"], b["batch"], perm, 4)
x = torch.randn(N, p.d_in, device=dev)
wpad = (len(p.fused_cols) + 15) // 16 * 16
h2p = ops.split_hidden(torch.randn(E, 32, device=dev))
w2p = torch.randn(32, wpad + 16, device=dev)

def run_fused(entries_np, label):
    ent = torch.from_numpy(np.ascontiguousarray(entries_np)).to(dev)
    ust = torch.from_numpy(mplan.fused_unit_map(entries_np)).to(dev)
    upt = ust.numel()
    asp = ops.split_a_tiles(w2p, entries_np) if os.environ.get('NO_ASPLIT') is None else None
    f = lambda: ops.tp_fused(x, h2p, w2p, geo["sh_sorted"], rowptr, src, ent, ust, upt, int(os.environ.get('LDS_PER_WAVE', p.fused_lds_floats_per_wave)), p.d_mid, 18.0, a_split=asp)
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    cols = sum(int(r[2]) * bin(int(r[4]) & 0xffffffff).count("1") for r in entries_np)
    print(f"fused {label:14s} entries {len(entries_np):2d} waves/tile {upt:3d} cols {cols:4d}  {dt*1e3:7.3f} ms", flush=True)

run_fused(p.group_entries, "all")
ge = p.group_entries
for kind in sorted(set(int(k) for k in ge[:, 0] if k >= 0)):
    sel = np.zeros(len(ge), dtype=bool)
    for i in range(len(ge)):
        if ge[i, 0] == kind:
            sel[i] = True
            if kind & mplan.TP_KIND_MERGED:
                sel[i + 1] = True          # the continuation record of a merged entry travels with it
    k = kind & (mplan.TP_KIND_MERGED - 1)
    run_fused(ge[sel], f"l1={k // mplan.TP_KIND_STRIDE} g={k % mplan.TP_KIND_STRIDE}{'m' if kind & mplan.TP_KIND_MERGED else ''}")
