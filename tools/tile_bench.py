#!/usr/bin/env python3
"""conv_tile vs the two-kernel conv per layer on the bench batch (HIP-event kernel times), over tile block sizes and
species mixes: where does the tile kernel's time go?   python3 tools/tile_bench.py [n_crystals]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd import ops
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.nn import conv as pconv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
graphs = synthetic.fcc64_graphs(B)
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
torch.manual_seed(35)
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
batch = collate(graphs, device=dev)
one = dict(batch)
one["atomic_numbers"] = torch.full_like(batch["atomic_numbers"], 29)


def run(tag, b, steps=6):
    with torch.no_grad():
        for _ in range(2):
            model(dict(b))
        torch.cuda.synchronize()
        ops.enable_event_timing(True, only=("tp_scatter", "conv_tile", "agg_linear"))
        t0 = time.perf_counter()
        for _ in range(steps):
            model(dict(b))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ev = {k: sum(v) / len(v) for k, v in ops.event_timings_ms().items()}
        ops.enable_event_timing(False)
    print(f"{tag:44s} fwd {dt*1e3:6.3f} ms | " + "  ".join(f"{k.split('/')[0][:9]}:{k.split('/')[1][:10]} {v:.3f}" for k, v in ev.items()), flush=True)


pconv.CONV_TILE = "1"
pconv.CONV_TILE_MIN_ROWS = 10**12
run("two-kernel, 10 species", batch)
run("two-kernel, 1 species", one)
pconv.CONV_TILE_MIN_ROWS = 0
for blk in [int(x) for x in os.environ.get("BLOCKS", "64,256,1024,2048,8192").split(",")]:
    pconv.CONV_TILE_BLOCK = blk
    run(f"tile, 1 species, block {blk}", one)
for blk in [int(x) for x in os.environ.get("BLOCKS10", "1024,2048,4096").split(",")]:
    pconv.CONV_TILE_BLOCK = blk
    run(f"tile, 10 species, block {blk}", batch)
