#!/usr/bin/env python3
"""Where a tp_fused wave's cycles go, per group kind (needs a `-DMATTEN_LAB -DTPF_TRACE` build: tools/tp_trace.sh).
Every contracting wave of the launch sums s_memtime differences at the phase boundaries of its chunk loop:
  prologue | per chunk: head (stage-load issue, B fragments from LDS, matrix phase, weight tile written)
           | contraction (weights / harmonics from LDS, CG code, neighbour gathers) | publish (wait for the stage loads,
             write them to LDS) | barrier | epilogue (agg stores, until vmcnt(0))
TARGET=view|full|l1|l2 selects the conv layer's plan (default: full last layer)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matten_amd import ops, plan as mplan, _lib
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
tgt = os.environ.get("TARGET", "full")
irr_in = irr
if tgt == "view":
    target = "32x0e+4x2e+2x4e"
elif tgt == "l1":   # second conv layer: input = what the first one can produce from 16x0e
    irr_in, target = "32x0e+16x1o+4x2e+2x3o+2x4e", irr
else:
    target = irr
p = mplan.plan_uvu(irr_in, Irreps.spherical_harmonics(4), target)
perm, rowptr, src, _ = ops.csr_build(b["edge_index"], N)
geo = ops.edge_geom(b["pos"], b["edge_index"], b["edge_cell_shift"], b["cell"], b["batch"], perm, 4)
x = torch.randn(N, p.d_in, device=dev)
wpad = (len(p.fused_cols) + 15) // 16 * 16
h2p = ops.split_hidden(torch.randn(E, 32, device=dev))
w2p = torch.randn(32, wpad + 16, device=dev)
lib = _lib.load()
lib.matten_lab_tp_trace.restype = ctypes.c_int
lib.matten_lab_tp_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]

NAMES = ["prologue", "head+mfma", "contract", "publish", "barrier", "epilogue", "total"]

def run(entries_np, label):
    ent = torch.from_numpy(np.ascontiguousarray(entries_np)).to(dev)
    um = mplan.fused_unit_map(entries_np)
    ust = torch.from_numpy(um).to(dev)
    upt = ust.numel()
    n_tiles = -(-N // 64)
    grid = -(-n_tiles // 8) * 8 * (-(-upt // 4))
    trace = torch.zeros(grid * 4, 16, dtype=torch.int32, device=dev)
    asp = ops.split_a_tiles(w2p, entries_np)
    f = lambda: ops.tp_fused(x, h2p, w2p, geo["sh_sorted"], rowptr, src, ent, ust, upt, p.fused_lds_floats_per_wave, p.d_mid, 18.0, a_split=asp)
    lib.matten_lab_tp_trace(None, 0, 0)
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); f(); ev1.record(); torch.cuda.synchronize()
    t_plain = ev0.elapsed_time(ev1)
    lib.matten_lab_tp_trace(trace.data_ptr(), grid * 4, -1)
    ev0.record(); f(); ev1.record(); torch.cuda.synchronize()
    lib.matten_lab_tp_trace(None, 0, 0)
    report(trace, f"{label}: {len(entries_np)} entries, {upt} units/tile, launch {t_plain:.3f} ms untraced / {ev0.elapsed_time(ev1):.3f} ms traced")


def report(trace, label):
    t = trace.cpu().numpy().astype(np.int64) & 0xffffffff
    t = t[t[:, 0] == 1]
    t0 = (t[:, 12] | (t[:, 13] << 32))
    span = (t0 + t[:, 10]).max() - t0.min()
    print(f"== {label}, {len(t)} traced waves, first wave in -> last wave out {span / 1e6:.3f} Mcycles")
    print(f"{'kind':>10s} {'lanes/node':>10s} {'TT':>3s} {'MT':>3s} {'pair':>4s} {'waves':>7s} {'chunks':>6s} | " + " ".join(f"{n:>10s}" for n in NAMES)
          + " | per chunk: " + " ".join(f"{n:>9s}" for n in NAMES[1:5]) + " | wave-cycle share")
    tot_all = t[:, 10].sum()
    keys = sorted(set(map(tuple, t[:, [1, 2]])))
    for k in keys:
        m = t[(t[:, 1] == k[0]) & (t[:, 2] == k[1])]
        kind, cul, paired, TT, MT = k[0] & 255, k[1] & 255, (k[1] >> 8) & 1, (k[1] >> 12) & 15, k[1] >> 16
        cols = m[:, [4, 5, 6, 7, 8, 9, 10]].mean(0)
        ch = m[:, 3].mean()
        per = m[:, 5:9].sum(0) / max(1, m[:, 3].sum())
        print(f"{'(%d,%d)%s' % (kind // 8, kind % 8, 'm' if k[0] & 256 else ''):>10s} {1 << cul:10d} {TT:3d} {MT:3d} {paired:4d} {len(m):7d} {ch:6.1f} | "
              + " ".join(f"{c:10.0f}" for c in cols) + " |            " + " ".join(f"{c:9.0f}" for c in per)
              + f" | {m[:, 10].sum() / tot_all:6.3f}"
              + f" | prologue up to its barrier {m[:, 14].mean():7.0f}, epilogue: next group's start {m[:, 15].mean():7.0f}")
    sums = t[:, [4, 5, 6, 7, 8, 9]].sum(0)
    print("all waves: share of wave cycles  " + "  ".join(f"{n} {s / tot_all:.3f}" for n, s in zip(NAMES[:6], sums)))

if tgt == "model":
    # the bench model itself (component-major neighbour sums, host-built A fragments, last layer as its dead-output view):
    # one traced launch per conv layer
    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
    torch.manual_seed(35)
    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            model(dict(b))
        for layer in range(4):
            trace = torch.zeros(1 << 20, 16, dtype=torch.int32, device=dev)
            lib.matten_lab_tp_trace(trace.data_ptr(), 1 << 20, layer)
            model(dict(b))
            torch.cuda.synchronize()
            lib.matten_lab_tp_trace(None, 0, 0)
            report(trace, f"bench model, conv layer {layer}")
    sys.exit(0)
run(p.group_entries, tgt)
if os.environ.get("PER_KIND"):
    ge = p.group_entries
    for kind in sorted(set(int(k) for k in ge[:, 0] if k >= 0)):
        sel = np.zeros(len(ge), dtype=bool)
        for i in range(len(ge)):
            if ge[i, 0] == kind:
                sel[i] = True
                if kind & mplan.TP_KIND_MERGED:
                    sel[i + 1] = True
        run(ge[sel], f"{tgt} kind ({(kind & 255) // 8},{kind % 8}) alone")
