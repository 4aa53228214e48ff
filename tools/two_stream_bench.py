#!/usr/bin/env python3
"""Does running two half batches on two HIP streams beat one full batch?  (VALU-bound tensor product of one half next to
the HBM-bound lin2 of the other.)  Crystals are independent, so the split is exact."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

dev = "cuda:0"
B = int(os.environ.get("B", 1000))
graphs = synthetic.fcc64_graphs(B)
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
torch.manual_seed(35)
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
full = collate(graphs, device=dev)
for parts in (1, 2, 4):
    n = B // parts
    batches = [collate(graphs[i * n:(i + 1) * n], device=dev) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def step(concurrent):
        with torch.no_grad():
            if not concurrent:
                return [model(dict(b))[0]["elastic_tensor_full"] for b in batches]
            outs = []
            cur = torch.cuda.current_stream()
            for b, s in zip(batches, streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(model(dict(b))[0]["elastic_tensor_full"])
            for s in streams:
                cur.wait_stream(s)
            return outs

    for conc in ((False,) if parts == 1 else (False, True)):
        for _ in range(5):
            step(conc)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            step(conc)
        torch.cuda.synchronize()
        print(f"parts {parts} {'concurrent streams' if conc else 'sequential'}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per {B} crystals", flush=True)
