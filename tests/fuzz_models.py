#!/usr/bin/env python3
"""Random small models (irreps, multiplicities, lmax, depth, normalisation, neighbour normalisation) on random small
crystals: product on the GPU against the oracle on the CPU -- the evaluation forward and, with a third argument "grad",
a training forward plus every parameter gradient of a random MSE loss (also randomises the activation type, the
normalisation method and the pooling).  Test infrastructure (it drives the oracle):
    tests/fuzz_models.py [n_cases] [seed] [grad]"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER, build_pair
from matten_amd.data.graph import collate, crystal_graph

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
GRAD = len(sys.argv) > 3 and sys.argv[3] == "grad"
if os.environ.get("NAN_EMPTY"):   # debugging aid: uninitialised output buffers hold NaN instead of whatever was there
    _empty, _empty_like = torch.empty, torch.empty_like
    def _nan_empty(*a, **k):
        t = _empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    def _nan_empty_like(*a, **k):
        t = _empty_like(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    torch.empty, torch.empty_like = _nan_empty, _nan_empty_like
bad = 0
for case in range(n_cases):
    lmax = int(rng.integers(1, 5))
    irr = []
    for l in range(lmax + 1):
        for p in "oe":
            if rng.random() < 0.75 or (l == 0 and p == "e"):
                irr.append(f"{int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 32]))}x{l}{p}")
    hp = dict(PAPER)
    hp["irreps_edge_sh"] = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lmax + 1))
    hp["conv_layer_irreps"] = "+".join(irr)
    hp["num_layers"] = int(rng.integers(1, 4))
    hp["species_embedding_dim"] = int(rng.choice([4, 16, 33]))
    hp["num_radial_basis"] = int(rng.choice([4, 8, 10]))
    hp["normalization"] = rng.choice(["batch", None])
    hp["average_num_neighbors"] = rng.choice(["auto", None])
    if GRAD:
        hp["normalization"] = rng.choice(["batch", "instance", None])
        hp["nonlinearity_type"] = rng.choice(["gate", "gate", "norm"])
        hp["reduce"] = rng.choice(["mean", "sum", "max"])
    # crystals: random triclinic cells with 1-6 atoms
    graphs, species = [], sorted(rng.choice(np.arange(1, 90), size=int(rng.integers(1, 5)), replace=False).tolist())
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 7))
        cell = 3.0 * np.eye(3) + rng.normal(0, 0.6, (3, 3))
        pos = rng.random((n, 3)) @ cell
        graphs.append(crystal_graph(pos, cell, rng.choice(species, size=n), 5.0))
    ds = {"allowed_species": species, "average_num_neighbors": float(np.mean([g["num_neigh"].mean() for g in graphs]))}
    if os.environ.get("CASE") and int(os.environ["CASE"]) != case:
        continue   # (the generator state above is consumed all the same)
    try:
        ref, model = build_pair(hp, ds, randomize_bn=True, seed=case)
    except (ValueError, RuntimeError, NotImplementedError) as e:  # e.g. no path to the gates: both sides must refuse
        print(case, "construction refused:", type(e).__name__, str(e)[:80])
        continue
    with torch.no_grad():
        want = ref.decode(collate(graphs))
        got = model.decode(collate(graphs, device="cuda:0"))["elastic_tensor_full"].cpu()
    scale = max(1e-6, want.abs().max().item())
    err = (got - want).abs().max().item() / scale
    flag = "" if err < 5e-4 else "  <-- BAD"
    gerr = g32 = 0.0
    if GRAD and len(graphs) > 1:   # (batch statistics need more than one sample to be meaningful)
        # the SAME oracle in fp64 anchors the comparison: instance normalisation of a crystal whose atoms are (nearly)
        # equivalent divides rounding noise by sqrt(eps), in the fp32 oracle as much as here -- such a case is judged by
        # how far the fp32 oracle itself is from fp64
        ref64 = copy.deepcopy(ref).double()
        b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in collate(graphs).items()}
        with torch.no_grad():
            want64 = ref64.decode(dict(b64))
        f32 = (want.double() - want64).abs().max().item() / scale
        fprod = (got.double() - want64).abs().max().item() / scale
        flag = "" if fprod < max(5e-4, 4 * f32) else "  <-- BAD"
        ref.train(), model.train(), ref64.train()
        tgt = torch.as_tensor(np.random.default_rng(1000 + case).normal(size=tuple(want.shape)), dtype=torch.float32)
        torch.nn.functional.mse_loss(ref.decode(collate(graphs)), tgt).backward()
        torch.nn.functional.mse_loss(ref64.decode(dict(b64)), tgt.double()).backward()
        torch.nn.functional.mse_loss(model.decode(collate(graphs, device="cuda:0"))["elastic_tensor_full"], tgt.cuda()).backward()
        named = dict(model.named_parameters())
        gscale = max([p.grad.abs().max().item() for p in ref64.parameters() if p.grad is not None] + [1e-12])
        rows = []
        for (k, p64), (_, p32) in zip(ref64.named_parameters(), ref.named_parameters()):
            if p64.grad is None:
                continue
            g = named[k].grad
            e32 = (p32.grad.double() - p64.grad).abs().max().item() / gscale
            ep = float("inf") if g is None else (g.cpu().double() - p64.grad).abs().max().item() / gscale
            g32, gerr = max(g32, e32), max(gerr, ep)
            rows.append((k, e32, ep))
        if os.environ.get("VERBOSE"):
            print("    crystal sizes", [int(g["pos"].shape[0]) for g in graphs])
            for k, e32, ep in rows:
                print(f"      {k:55s} fp32 oracle {e32:.1e}  product {ep:.1e}   (of the largest fp64 gradient)")
        if not gerr < max(5e-3, 4 * g32):
            flag += "  <-- BAD (gradients)"
        err = fprod
    bad += bool(flag)
    print(f"{case:3d} lmax {lmax} layers {hp['num_layers']} {str(hp.get('nonlinearity_type'))[:4]}/{str(hp['normalization'])[:5]:5s} "
          f"irreps {hp['conv_layer_irreps'][:48]:48s} rel err {err:.1e} grad {gerr:.1e} (fp32 oracle {g32:.1e}){flag}")
print("bad", bad)
sys.exit(1 if bad else 0)
