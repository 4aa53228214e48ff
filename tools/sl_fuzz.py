#!/usr/bin/env python3
"""Randomised check of matten_species_linear against a dense fp64 evaluation: random irreps (l <= 4), multiplicities,
species counts, row counts, with / without addend.  Usage: sl_fuzz.py [n_cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matten_amd import ops, plan as mplan

dev = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
gen = torch.Generator(device=dev).manual_seed(1)
worst, bad = 0.0, 0
for case in range(n_cases):
    ls = sorted(set(rng.integers(0, 5, size=rng.integers(1, 5)).tolist()))
    par = lambda l: "e" if rng.random() < 0.5 else "o"
    irs = [(l, par(l)) for l in ls]
    big = rng.random() < 0.15
    iin = "+".join(f"{int(rng.integers(1, 400 if big else 40))}x{l}{p}" for l, p in irs)
    iout = "+".join(f"{int(rng.integers(1, 100 if rng.random() < 0.2 else 20))}x{l}{p}" for l, p in irs)
    S = int(rng.integers(1, 13)); N = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 200, 1000, 3000])); add = bool(rng.integers(0, 2))
    lp = mplan.plan_fctp(iin, S, iout)
    x = torch.randn(N, lp.d_in, device=dev, generator=gen); wp = torch.randn(S, lp.w_stride, device=dev, generator=gen)
    a = torch.randn(N, lp.d_out, device=dev, generator=gen) if add else None
    sp = torch.randint(0, S, (N,), device=dev, generator=gen)
    order, seg, _, _ = ops.csr_build(torch.stack([torch.arange(N, device=dev), sp]), S)
    items = [torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in lp.passes]
    got = ops.species_linear(x, (order, seg), wp, lp.w_stride, items, lp.d_out, a, lp.fully_covered)
    want = a.double().clone() if add else torch.zeros(N, lp.d_out, dtype=torch.float64, device=dev)
    for p in lp.passes:
        for (xo, d, mi, wo, mo, oo, _, _) in p.tolist():
            W = wp.double()[sp][:, wo:wo + mi * mo].reshape(N, mi, mo)
            X = x.double()[:, xo:xo + mi * d].reshape(N, mi, d)
            want[:, oo:oo + mo * d] += torch.einsum("nuv,num->nvm", W, X).reshape(N, mo * d)
    rel = (got.double() - want).abs().max().item() / max(1e-6, want.abs().max().item())
    worst = max(worst, rel)
    if not rel < 3e-6:
        bad += 1
        print("BAD", iin, "->", iout, "S", S, "N", N, "add", add, "rel", rel)
print(f"{n_cases} cases, worst rel err {worst:.2e}, bad {bad}")
sys.exit(1 if bad else 0)
