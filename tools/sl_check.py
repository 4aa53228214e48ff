"""species_linear against a dense fp64 evaluation of the same segment tables (debug / regression helper)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from matten_amd import ops, plan as mplan
from matten_amd.o3 import Irreps

dev = "cuda:0"
torch.manual_seed(0)


def check(irreps_in, irreps_out, S, N, with_add):
    lp = mplan.plan_fctp(irreps_in, S, irreps_out)
    x = torch.randn(N, lp.d_in, device=dev)
    wp = torch.randn(S, lp.w_stride, device=dev)
    add = torch.randn(N, lp.d_out, device=dev) if with_add else None
    species = torch.randint(0, S, (N,), device=dev)
    order, seg, _, _ = ops.csr_build(torch.stack([torch.arange(N, device=dev), species]), S)
    items = [torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in lp.passes]
    got = ops.species_linear(x, (order, seg), wp, lp.w_stride, items, lp.d_out, add, lp.fully_covered)
    want = add.double().clone() if with_add else torch.zeros(N, lp.d_out, dtype=torch.float64, device=dev)
    xd, wd = x.double(), wp.double()
    for p in lp.passes:
        for (xo, d, mi, wo, mo, oo, _, _) in p.tolist():
            W = wd[species][:, wo:wo + mi * mo].reshape(N, mi, mo)
            X = xd[:, xo:xo + mi * d].reshape(N, mi, d)
            want[:, oo:oo + mo * d] += torch.einsum("nuv,num->nvm", W, X).reshape(N, mo * d)
    err = (got.double() - want).abs()
    rel = err.max().item() / want.abs().max().item()
    bad = (err > 1e-4 * want.abs().max()).nonzero()
    if bad.shape[0] or rel > 1e-5:
        print(f"{irreps_in} -> {irreps_out} N={N} add={with_add}: rel err {rel:.2e}; bad entries {bad.shape[0]}",
              (bad[:5].tolist(), sorted(set(bad[:, 1].tolist()))[:20]) if bad.shape[0] else "")
    return rel


def model_cases():
    sys.path.insert(0, "tests")
    from common import PAPER
    from matten_amd.data import synthetic
    from matten_amd.model_factory.tfn_scalar_tensor import create_model
    m = create_model(dict(PAPER), {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0})
    for name, mod in m.named_modules():
        if type(mod).__name__ == "SpeciesLinear" and mod.n_species is not None:
            yield name, str(mod.irreps_in), str(mod.irreps_out), mod.n_species


worst = 0
for name, iin, iout, S in model_cases():
    print(name, iin, "->", iout)
    for N in (192, 200):
        for with_add in (False, True):
            worst = max(worst, check(iin, iout, S, N, with_add))
irr = "32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e"
mid = mplan.plan_uvu(irr, Irreps.spherical_harmonics(4), irr).irreps_out
for N in (1, 17, 64, 1000, 4097):
    for with_add in (False, True):
        worst = max(worst, check(mid, irr, 10, N, with_add))
        worst = max(worst, check("200x0e+37x1o+5x2e", "40x0e+17x1o+33x2e", 3, N, with_add))
        worst = max(worst, check("16x0e", "8x0e+3x1o", 2, N, with_add))
for mo in list(range(1, 100)) + [127, 128, 129, 160, 161]:
    for d, mi in ((0, 8), (1, 5), (2, 3)):
        par = "e" if d % 2 == 0 else "o"
        worst = max(worst, check(f"{mi}x{d}{par}", f"{mo}x{d}{par}", 2, 37, True))
for mi in list(range(1, 40)) + [159, 160, 161, 170, 321]:
    for d in range(5):
        par = "e" if d % 2 == 0 else "o"
        worst = max(worst, check(f"{mi}x{d}{par}+3x0e", f"5x{d}{par}+2x0e", 3, 50, bool(mi % 2)))
print("WORST", worst)
sys.exit(0 if worst < 1e-5 else 1)
