"""Shared hyper-parameter sets and builders for the tests (hparams follow the reference's files)."""
import copy

import torch

# reference pretrained/20230627/config_final.yaml:24-42
PAPER = {
    "species_embedding_dim": 16,
    "irreps_edge_sh": "0e + 1o + 2e + 3o + 4e",
    "num_radial_basis": 8,
    "radial_basis_start": 0.0,
    "radial_basis_end": 5.0,
    "radial_basis_type": "bessel",
    "num_layers": 3,
    "invariant_layers": 2,
    "invariant_neurons": 32,
    "average_num_neighbors": "auto",
    "conv_layer_irreps": "32x0o+32x0e + 16x1o+16x1e + 4x2o+4x2e + 2x3o+2x3e + 2x4e",
    "nonlinearity_type": "gate",
    "normalization": "batch",
    "resnet": True,
    "conv_to_output_hidden_irreps_out": "16x0e + 2x2e + 4e",
    "output_format": "irreps",
    "output_formula": "ijkl=jikl=klij",
    "reduce": "mean",
}

# reference tests/model/test_tfn_tensor.py:23-42
EQUIV_TEST = {
    "species_embedding_dim": 32,
    "irreps_edge_sh": "0e + 1o + 2e + 3o + 4e",
    "num_radial_basis": 10,
    "radial_basis_start": 0.0,
    "radial_basis_end": 5.0,
    "radial_basis_type": "bessel",
    "num_layers": 3,
    "invariant_layers": 2,
    "invariant_neurons": 32,
    "average_num_neighbors": None,
    "conv_layer_irreps": "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e+4x3o+4x3e+4x4o+4x4e",
    "nonlinearity_type": "gate",
    "normalization": None,
    "resnet": True,
    "conv_to_output_hidden_irreps_out": "2x0e + 2x2e + 4e",
    "output_format": "cartesian",
    "output_formula": "ijkl=jikl=klij",
    "reduce": "mean",
}

# lmax=2 variant of config 4 (reference scripts/configs/atomic_tensor.yaml:29,54 grafted on the elasticity head)
LMAX2 = dict(
    PAPER,
    irreps_edge_sh="0e + 1o + 2e",
    conv_layer_irreps="32x0o+32x0e+16x1o+16x1e+4x2o+4x2e",
)


def build_pair(hparams, dataset_hparams, seed=35, device="cuda:0", randomize_bn=False, atomic=False):
    """(oracle on CPU, product on `device`) sharing one state_dict."""
    from matten_amd.model_factory.tfn_atomic_tensor import AtomicTensorModel
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
    from oracle.matten_ref.model import AtomicTensorOracle, ScalarTensorOracle

    torch.manual_seed(seed)
    ref = (AtomicTensorOracle if atomic else ScalarTensorOracle)(copy.deepcopy(hparams), dataset_hparams).eval()
    if randomize_bn:
        g = torch.Generator().manual_seed(seed + 1)
        for name, buf in ref.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            if name.endswith("running_var"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        for name, p in ref.named_parameters():
            if ".norm.n.weight" in name:
                p.data.copy_(0.5 + torch.rand(p.shape, generator=g))
            if ".norm.n.bias" in name:
                p.data.copy_(0.1 * torch.randn(p.shape, generator=g))
    model = (AtomicTensorModel if atomic else ScalarTensorModel)(
        tasks="nmr_tensor" if atomic else None, backbone_hparams=copy.deepcopy(hparams), dataset_hparams=dataset_hparams)
    missing, unexpected = model.load_state_dict(ref.state_dict(), strict=False)
    assert not missing, missing
    assert all(k.endswith("output_mask") or k.endswith("tp.tp.weight") for k in unexpected), unexpected
    if device is not None:
        model = model.to(device)
    return ref, model.eval()


# reference scripts/configs/atomic_tensor.yaml:20-72 (per-atom symmetric 2-tensor, e.g. NMR shielding)
ATOMIC = {
    "species_embedding_dim": 16,
    "irreps_edge_sh": "0e + 1o + 2e",
    "num_radial_basis": 8,
    "radial_basis_start": 0.0,
    "radial_basis_end": 5.0,
    "radial_basis_type": "bessel",
    "num_layers": 3,
    "invariant_layers": 2,
    "invariant_neurons": 32,
    "average_num_neighbors": "auto",
    "conv_layer_irreps": "32x0o+32x0e + 16x1o+16x1e + 4x2o+4x2e",
    "nonlinearity_type": "gate",
    "normalization": "batch",
    "resnet": True,
    "output_format": "irreps",
    "output_formula": "ij=ji",
    "reduce": "mean",
}
