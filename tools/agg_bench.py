#!/usr/bin/env python3
"""lin2 alone on the four conv layers of the paper model, 1000 fcc-64 crystals (64 000 rows, 10 species): the
component-major stream (matten_agg_linear) next to the mul_ir segment-table kernels, with the bytes each moves."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import ops
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

N = int(os.environ.get("ROWS", "64000"))
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
torch.manual_seed(0)
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
species = torch.randint(0, 10, (N,), device="cuda:0")
if os.environ.get("SORTED") == "1":   # rows already grouped: the order indirection becomes the identity
    species = torch.sort(species).values
order, seg, _ = ops.group_by_key(species, 10)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for m in [m for m in model.modules() if type(m).__name__ == "PointConv"]:
    ap, lp = m.agg_plan, m.lin2.plan
    add = torch.randn(N, lp.d_out, device="cuda:0")
    line = f"d_mid {m.tp.plan.d_mid:5d}"
    if ap is not None:
        agg = torch.randn(N, ap.ld, device="cuda:0")
        t, dev = m._agg_tables, agg.device
        wtab = m._agg_wtab.get(m.lin2.weight)
        dt = timed(lambda: ops.agg_linear(agg, (order, seg), wtab, t.get("io", dev), t.get("blocks", dev), ap.d_out, add=add))
        gb = N * (ap.ld + 2 * lp.d_out) * 4 / 1e9
        line += f" | agg_linear ld {ap.ld:5d}: {dt*1e3:.3f} ms  {gb/dt/1e3:.2f} TB/s"
        dt = timed(lambda: ops.agg_linear(agg, (order, seg), wtab, t.get("io", dev), t.get("blocks", dev), ap.d_out))
        line += f" (no addend {dt*1e3:.3f})"
    aggo = torch.randn(N, m.tp.plan.d_mid, device="cuda:0")
    dt = timed(lambda: m.lin2(aggo, (order, seg), add=add))
    gb = N * (m.tp.plan.d_mid + 2 * lp.d_out) * 4 / 1e9
    line += f" | mul_ir kernel: {dt*1e3:.3f} ms  {gb/dt/1e3:.2f} TB/s"
    print(line, flush=True)
