#!/usr/bin/env python3
"""csr_build / group_by_key under hipGraph capture + replay (which of the two breaks a captured step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matten_amd import ops

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "csr"
N, E = 3000, 50000
g = torch.Generator().manual_seed(1)
ei = torch.randint(0, N, (2, E), generator=g).to(dev)
key = torch.randint(0, 10, (N,), generator=g).to(dev)
fn = (lambda: ops.csr_build(ei, N)) if which == "csr" else (lambda: ops.group_by_key(key, 10))
want = [t.clone() for t in fn()]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        fn()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = fn()
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    graph.replay()
    torch.cuda.synchronize()
    print("replay", i, [bool(torch.equal(a, b)) for a, b in zip(out, want)], flush=True)
