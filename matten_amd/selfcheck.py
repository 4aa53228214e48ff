"""
Load-time self-check of the kernels whose correctness hangs on a compiler flag.

``species_linear.hip`` / ``species_linear_rows.hip`` issue their matrix instructions through inline asm (the compiler's
hazard recogniser does not see them) and are only correct when no accumulator lives in scratch memory, which the
Makefile guarantees with ``-mllvm -simplifycfg-sink-common=false`` (csrc/Makefile).  A toolchain
bump or a hand-made build without that flag could silently bring back accumulators with a stale half.  So the first
species-linear call on a device runs both kernels on a handful of small shapes -- the family that exposed the
miscompile (d = 1 blocks with an addend, output multiplicities across the 16-channel tile boundaries) plus wider irreps
-- with SMALL-INTEGER inputs and weights: every product and partial sum is an integer below 2^24, so the fp32 result is
exact whatever the summation order, and it is compared BIT FOR BIT with an int64 evaluation on the host.  On a mismatch
the library refuses to run (``MattenHipError``).  One-off cost: ~20 tiny launches and one host sync per process and device.
``MATTEN_SELFCHECK=0`` skips it (profiling runs that must not see the extra launches).
"""
import os
from typing import Dict

import numpy as np
import torch

from . import _lib

_done: Dict[int, bool] = {}

# (irreps_in, irreps_out, species, rows): d = 1 blocks with an addend around the tile boundaries first
_SHAPES = [
    ("8x0e", "5x0e", 2, 37), ("8x0e", "17x0e", 2, 37), ("8x0e", "33x0e", 2, 37), ("8x0e", "78x0e", 2, 37),
    ("40x0e", "144x0e", 2, 37), ("5x1o", "19x1o", 2, 37), ("3x2e", "35x2e", 2, 37),
    ("78x0e+16x1o+4x2e+2x3o+2x4e", "32x0e+16x1o+4x2e+2x3o+2x4e", 3, 70),
    ("170x0e+33x1o", "65x0e+17x1o", 2, 50),
]


def _case(ops, plan_mod, irreps_in, irreps_out, S, N, dev, gen, variant):
    lp = plan_mod.plan_fctp(irreps_in, S, irreps_out)
    x = torch.randint(-2, 3, (N, lp.d_in), generator=gen).float()
    wp = torch.randint(-1, 2, (S, lp.w_stride), generator=gen).float()
    add = torch.randint(-3, 4, (N, lp.d_out), generator=gen).float()
    species = torch.randint(0, S, (N,), generator=gen)
    species[: N // 3] = 0
    order = torch.argsort(species, stable=True).to(torch.int32)
    seg = torch.zeros(S + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(torch.bincount(species, minlength=S), 0).to(torch.int32)
    want = add.to(torch.int64).clone()
    xi, wi = x.to(torch.int64), wp.to(torch.int64)
    for p in lp.passes:
        for (xo, d, mi, wo, mo, oo, _, _) in p.tolist():
            if mi == 0:
                continue
            W = wi[species][:, wo:wo + mi * mo].reshape(N, mi, mo)
            X = xi[:, xo:xo + mi * d].reshape(N, mi, d)
            want[:, oo:oo + mo * d] += torch.einsum("nuv,num->nvm", W, X).reshape(N, mo * d)
    items = [torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in lp.passes]
    got = ops.species_linear(x.to(dev), (order.to(dev), seg.to(dev)), wp.to(dev), lp.w_stride, items, lp.d_out,
                             add.to(dev), lp.fully_covered, variant=variant)
    return got, want


def check_species_linear(device) -> None:
    """run once per process and device; raises MattenHipError when a kernel's result is not exact"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if _done.get(idx) or os.environ.get("MATTEN_SELFCHECK", "1") == "0":
        return
    _done[idx] = True      # set first: the check itself calls ops.species_linear
    if torch.cuda.is_current_stream_capturing():
        _done[idx] = False  # a host sync cannot be captured; check at the next eager call
        return
    from . import ops, plan as plan_mod

    gen = torch.Generator().manual_seed(20250711)
    pending = []
    for variant in ("rows", "stream"):
        for (iin, iout, S, N) in _SHAPES:
            got, want = _case(ops, plan_mod, iin, iout, S, N, device, gen, variant)
            pending.append((variant, iin, iout, got, want))
    bad = []
    for variant, iin, iout, got, want in pending:   # one sync for all of them
        g = got.cpu()
        if not torch.equal(g, want.float()):
            n_bad = int((g != want.float()).sum())
            bad.append(f"{variant}: {iin} -> {iout}: {n_bad} of {g.numel()} outputs wrong")
    if bad:
        _done[idx] = False
        raise _lib.MattenHipError(
            "species-linear self-check FAILED (integer inputs, exact expected result): " + "; ".join(bad[:6]) +
            ".  libmatten_hip.so was probably built without `-mllvm -simplifycfg-sink-common=false` for "
            "species_linear*.hip (csrc/Makefile SL_FLAGS) or with a toolchain that schedules around the inline-asm "
            "matrix instructions differently; rebuild with `make -C matten_amd/csrc clean all`.")
