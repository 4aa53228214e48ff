"""Graph construction: host KD-tree builder vs the device neighbour list (section 8(f)-1), fcc-64 crystals."""
import sys
import time

import torch

sys.path.insert(0, ".")
from matten_amd import ops  # noqa: E402
from matten_amd.data import synthetic  # noqa: E402
from matten_amd.data.graph import batch_graphs_gpu, collate, crystal_graph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
structs = synthetic.fcc64_structures(n)
triples = [(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in structs]
t0 = time.perf_counter()
graphs = [crystal_graph(*t, 5.0) for t in triples[:100]]
collate(graphs, device="cuda")
torch.cuda.synchronize()
host = (time.perf_counter() - t0) / 100
batch_graphs_gpu(triples[:10], 5.0, "cuda")
torch.cuda.synchronize()
ops.enable_event_timing(True)
t0 = time.perf_counter()
b = batch_graphs_gpu(triples, 5.0, "cuda")
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / n
ev = {k: sum(v) for k, v in ops.event_timings_ms().items()}
import numpy as np
from matten_amd.data.graph import batch_graphs_gpu_soa
sizes = np.array([len(t[0]) for t in triples]); ptr = np.concatenate([[0], np.cumsum(sizes)])
pos = np.concatenate([t[0] for t in triples]); cell = np.stack([t[1] for t in triples]); Z = np.concatenate([t[2] for t in triples])
batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): batch_graphs_gpu_soa(pos, cell, Z, ptr, 5.0, "cuda")
torch.cuda.synchronize()
print(f"struct-of-arrays entry: {(time.perf_counter() - t0) / 5 / n * 1e3:.4f} ms/crystal end to end")
print(f"host builder {host*1e3:.3f} ms/crystal; device builder {gpu*1e3:.4f} ms/crystal end to end "
      f"({n} crystals, E={b['edge_index'].shape[1]}), kernels: {ev}")
