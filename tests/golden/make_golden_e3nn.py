#!/usr/bin/env python3
"""
Pins the CPU oracle to the REAL e3nn: writes tests/golden/e3nn_golden.npz.

Run in an environment that has ``e3nn==0.5.1`` (the reference's pin, pyproject.toml:29), ``torch_scatter``,
``torch_geometric``, ``pytorch_lightning``, ``pymatgen``, ``ase`` -- i.e. where the reference itself runs -- with the
reference checkout next to this repository or at $MATTEN_REFERENCE (default /root/reference):

    python tests/golden/make_golden_e3nn.py

Neither the build container nor the MI355X box has e3nn (no network, no wheel): there the script stops with the
message below, `tests/test_oracle.py::test_oracle_matches_e3nn_golden` reports "parity unpinned", and DESIGN.md says so.
Only the .npz travels; nothing of the reference or of e3nn is copied into it but numbers:

  w3j_{l1}_{l2}_{l3}        o3.wigner_3j for every l1, l2, l3 <= 4 with |l1-l2| <= l3 <= l1+l2 (odd-sum triples included)
  sh_points / sh_values     o3.spherical_harmonics([0..4], x, normalize=True, normalization="component")
  cart_basis_ijkl / _ij     CartesianTensor(formula).reduced_tensor_products().change_of_basis
  normalize2mom_*           e3nn.math.normalize2mom(act).cst for silu, sigmoid, tanh, abs and the shifted softplus
  soft_one_hot_*            e3nn.math.soft_one_hot_linspace(r, 0, 5, 8, "bessel", cutoff=True)
  teo/*                     the TeO fixture (reference tests/test_files/elastic_tensor_one.json) through the reference's
                            own data pipeline and create_model with the hparams and seed of
                            tests/model/test_tfn_tensor.py:23-42,99: the batch dict, the model's state_dict, the
                            node features after every backbone module and the final Cartesian tensor
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MATTEN_REFERENCE", "/root/reference")

try:
    import e3nn
    from e3nn import o3
    from e3nn.io import CartesianTensor
    from e3nn.math import normalize2mom, soft_one_hot_linspace
except ImportError as exc:  # the expected outcome on the build and GPU boxes
    sys.exit(f"make_golden_e3nn: e3nn is not importable here ({exc}); parity with real e3nn stays UNPINNED. "
             "Run this script where the reference runs (e3nn==0.5.1) and commit tests/golden/e3nn_golden.npz.")

import numpy as np
import torch

sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "tests", "model"))


def main():
    if e3nn.__version__ != "0.5.1":
        print(f"warning: e3nn {e3nn.__version__}, the reference pins 0.5.1", file=sys.stderr)
    out = {"e3nn_version": np.array(e3nn.__version__)}
    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1):
                out[f"w3j_{l1}_{l2}_{l3}"] = o3.wigner_3j(l1, l2, l3, dtype=torch.float64).numpy()
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(64, 3, generator=g, dtype=torch.float64)
    out["sh_points"] = pts.numpy()
    out["sh_values"] = o3.spherical_harmonics([0, 1, 2, 3, 4], pts, True, "component").numpy()
    for name, formula in (("ijkl", "ijkl=jikl=klij"), ("ij", "ij=ji")):
        ct = CartesianTensor(formula)
        out[f"cart_basis_{name}"] = ct.reduced_tensor_products().change_of_basis.double().numpy()
        out[f"cart_irreps_{name}"] = np.array(str(ct))
    acts = {"silu": torch.nn.functional.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "abs": torch.abs,
            "ssp": lambda x: torch.nn.functional.softplus(x) - float(np.log(2.0))}
    for k, f in acts.items():
        out[f"normalize2mom_{k}"] = np.array(normalize2mom(f).cst)
    r = torch.linspace(0.05, 5.5, 110, dtype=torch.float64)   # not 0: sin(x)/x is 0/0 there in e3nn too
    out["soft_one_hot_r"] = r.numpy()
    out["soft_one_hot_bessel"] = soft_one_hot_linspace(r, 0.0, 5.0, 8, basis="bessel", cutoff=True).numpy()

    # ---- the reference's own equivariance-test setup (tests/model/test_tfn_tensor.py) ----
    import pytorch_lightning
    from test_tfn_tensor import get_model, load_dataset   # reference test helpers: hparams :23-42, data module :52-69

    from matten.utils import ToCartesian

    pytorch_lightning.seed_everything(35)
    model = get_model("cartesian", "ijkl=jikl=klij").eval()
    loader = load_dataset(os.path.join(REF, "tests", "test_files", "elastic_tensor_one.json"), root="/tmp")
    batch = next(iter(loader))
    graphs = batch.tensor_property_to_dict()
    for k, v in graphs.items():
        if isinstance(v, torch.Tensor):
            out[f"teo/in/{k}"] = v.detach().numpy()
    for k, v in model.state_dict().items():
        out[f"teo/state/{k}"] = v.detach().numpy()
    acts_out = {}

    def hook(name):
        def fn(mod, inp, res):
            if isinstance(res, dict) and "node_features" in res:
                acts_out[name] = res["node_features"].detach().clone()
        return fn

    handles = [m.register_forward_hook(hook(n)) for n, m in model.named_children()]
    with torch.no_grad():
        res = model(dict(graphs))
        cart = ToCartesian("ijkl=jikl=klij")(res["my_model_output"])
    for h in handles:
        h.remove()
    for name, t in acts_out.items():
        out[f"teo/act/{name}"] = t.numpy()
    out["teo/my_model_output"] = res["my_model_output"].numpy()
    out["teo/cartesian"] = cart.numpy()
    np.savez_compressed(os.path.join(HERE, "e3nn_golden.npz"), **out)
    print(f"wrote {os.path.join(HERE, 'e3nn_golden.npz')}: {len(out)} arrays")


if __name__ == "__main__":
    main()
