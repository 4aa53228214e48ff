#!/usr/bin/env python3
"""Micro-benchmark of species_linear on a short-row shape (lin1 of the last conv layer, 246 -> 246)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matten_amd import ops, plan as mplan
dev = "cuda:0"
N, S = 64000, 10
irr = "32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e"
lp = mplan.plan_fctp(irr, S, irr)
x = torch.randn(N, lp.d_in, device=dev)
wp = torch.randn(S, lp.w_stride, device=dev)
items = [torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in lp.passes]
species = torch.randint(0, S, (N,), device=dev)
order, seg, _, _ = ops.csr_build(torch.stack([torch.arange(N, device=dev), species]), S)
f = lambda: ops.species_linear(x, (order, seg), wp, lp.w_stride, items, lp.d_out, None, True)
for _ in range(3): f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): f()
torch.cuda.synchronize(); print(f"lin1 {lp.d_in}->{lp.d_out}, {N} rows: {(time.perf_counter()-t)/20*1e6:.1f} us  (w_stride {lp.w_stride})")
