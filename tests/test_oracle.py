"""
CPU tests of the oracle: the reference's own pins for this path plus the
mathematical properties that fix the e3nn conventions, plus the committed golden vectors.
"""
import math
import os

import numpy as np
import pytest
import torch

from common import EQUIV_TEST, PAPER
from oracle.e3nn_lite import io, nn as enn, o3
from oracle.e3nn_lite.math import soft_one_hot_linspace
from oracle.e3nn_lite.scatter import scatter
from oracle.matten_ref import data as rdata
from oracle.matten_ref import nn as rnn
from oracle.matten_ref.model import ScalarTensorOracle, ToCartesian, create_model


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_golden.npz"))


# ---- reference tests/nn/test_embedding.py:6-13 ----------------------------------------------
def test_atomic_number_to_index_known_answer():
    n2i = rnn._AtomicNumberToIndex([6, 1, 8])
    index = n2i(torch.tensor([6, 6, 8, 1, 8]))
    assert index.dtype == torch.long
    assert torch.equal(index, torch.tensor([1, 1, 2, 0, 2]))
    with pytest.raises(RuntimeError, match="Invalid atomic numbers"):
        n2i(torch.tensor([6, 9]))
    with pytest.raises(RuntimeError, match="got invalid atomic numbers `7`"):
        n2i(torch.tensor([6, 7]))


# ---- reference tests/model/test_tfn_tensor.py:98-139 ----------------------------------------
def test_model_equivariance_and_symmetry(golden_dir):
    s = rdata.structures_from_json(os.path.join(golden_dir, "elastic_tensor_one.json"))[0]
    torch.manual_seed(35)
    model = create_model(dict(EQUIV_TEST), {"allowed_species": [8, 52]}).eval()
    torch.manual_seed(35)
    Q = o3.rand_matrix().double()
    g1 = rdata.crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0)
    g2 = rdata.crystal_graph(s["cart_coords"] @ Q.numpy().T, s["lattice"] @ Q.numpy().T, s["atomic_numbers"], 5.0)
    assert g1["pos"].shape[0] == 8 and g1["edge_index"].shape[1] == 252
    tc = ToCartesian("ijkl=jikl=klij")
    with torch.no_grad():
        pred = tc(model(rdata.collate([g1]))["my_model_output"])[0]
        pred_rot = tc(model(rdata.collate([g2]))["my_model_output"])[0]
    assert torch.allclose(pred, torch.swapaxes(pred, 0, 1))
    assert torch.allclose(pred, torch.swapaxes(pred, 2, 3))
    assert torch.allclose(pred, torch.swapaxes(torch.swapaxes(pred, 0, 2), 1, 3))
    Qf = Q.float()
    x = torch.einsum("im, jn, kp, lq, mnpq -> ijkl", Qf, Qf, Qf, Qf, pred)
    assert torch.allclose(x, pred_rot, atol=1e-4)


# ---- conventions -----------------------------------------------------------------------------
def test_irreps_sort_simplify_like_e3nn():
    ir = o3.Irreps("32x0o+32x0e + 16x1o+16x1e + 4x2o+4x2e + 2x3o+2x3e + 2x4e")
    assert str(ir) == "32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e" and ir.dim == 246
    mixed = o3.Irreps("2x1o+3x0e+1x0o+4x0e")
    srt, p, inv = mixed.sort()
    assert str(srt) == "1x0o+3x0e+4x0e+2x1o" and str(srt.simplify()) == "1x0o+7x0e+2x1o"
    assert p == (3, 1, 0, 2) and inv == (2, 1, 3, 0)
    assert [str(i) for i in o3.Irrep("1o") * o3.Irrep("2e")] == ["1o", "2o", "3o"]
    assert (o3.Irrep("0e") == o3.Irreps("0e")) is False  # reference nn/utils.py:210: dead clause


def test_wigner_3j_census_and_anchors(golden):
    dense = nnz = 0
    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1):
                C = o3.wigner_3j(l1, l2, l3, dtype=torch.float64)
                assert abs(C.norm().item() - 1) < 1e-12
                dense += C.numel()
                nnz += int((C.abs() > 1e-12).sum())
    assert (dense, nnz) == (13075, 2052)  # SURVEY.md A.2 census
    for l in range(5):
        eye = torch.eye(2 * l + 1, dtype=torch.float64) / math.sqrt(2 * l + 1)
        assert torch.allclose(o3.wigner_3j(l, 0, l, dtype=torch.float64)[:, 0, :], eye)
        assert torch.allclose(o3.wigner_3j(0, l, l, dtype=torch.float64)[0], eye)
        assert torch.allclose(o3.wigner_3j(l, l, 0, dtype=torch.float64)[:, :, 0], eye)
    assert abs(o3.wigner_3j(1, 1, 1, dtype=torch.float64)[0, 1, 2].item() - 1 / math.sqrt(6)) < 1e-12
    for k in ("w3j_111", "w3j_224", "w3j_444"):
        l1, l2, l3 = (int(c) for c in k[-3:])
        assert np.allclose(o3.wigner_3j(l1, l2, l3, dtype=torch.float64).numpy(), golden[k], atol=1e-14)


def test_spherical_harmonics_closed_forms_and_golden(golden):
    g = torch.Generator().manual_seed(0)
    v = torch.randn(50, 3, dtype=torch.float64, generator=g)
    n = v / v.norm(dim=1, keepdim=True)
    x, y, z = n[:, 0], n[:, 1], n[:, 2]
    Y = o3.spherical_harmonics([0, 1, 2, 3, 4], v, True, "norm")
    s = math.sqrt
    l2 = torch.stack([s(3) * x * z, s(3) * x * y, y * y - 0.5 * (x * x + z * z), s(3) * y * z,
                      s(3) / 2 * (z * z - x * x)], 1)
    l3 = torch.stack([s(5 / 8) * x * (3 * z * z - x * x), s(15) * x * y * z, s(3 / 8) * x * (4 * y * y - x * x - z * z),
                      0.5 * y * (2 * y * y - 3 * x * x - 3 * z * z), s(3 / 8) * z * (4 * y * y - x * x - z * z),
                      s(15) / 2 * y * (z * z - x * x), s(5 / 8) * z * (z * z - 3 * x * x)], 1)
    assert torch.allclose(Y[:, :1], torch.ones(50, 1, dtype=torch.float64))
    assert torch.allclose(Y[:, 1:4], n)
    assert torch.allclose(Y[:, 4:9], l2, atol=1e-13)
    assert torch.allclose(Y[:, 9:16], l3, atol=1e-13)
    Yc = o3.spherical_harmonics([0, 1, 2, 3, 4], v, True, "component")
    for l in range(5):
        assert torch.allclose((Yc[:, l * l:(l + 1) ** 2] ** 2).sum(1), torch.full((50,), 2.0 * l + 1, dtype=torch.float64))
    got = o3.spherical_harmonics([0, 1, 2, 3, 4], torch.from_numpy(golden["sh_points"]), True, "component")
    assert np.allclose(got.numpy(), golden["sh_values"], atol=1e-14)


def test_cg_is_equivariant_under_sh_wigner_d():
    """C_{ijk} D1_{ii'} D2_{jj'} D3_{kk'} = C_{i'j'k'} with D from the spherical harmonics themselves."""
    R = o3.rand_matrix(torch.Generator().manual_seed(7))
    D = [o3.wigner_D_from_sh(l, R) for l in range(5)]
    for l in range(5):
        assert torch.allclose(D[l] @ D[l].T, torch.eye(2 * l + 1, dtype=torch.float64), atol=1e-9)
    for (l1, l2, l3) in [(1, 1, 1), (1, 1, 2), (2, 2, 2), (2, 3, 4), (4, 4, 4), (3, 4, 2), (1, 2, 3)]:
        C = o3.wigner_3j(l1, l2, l3, dtype=torch.float64)
        C2 = torch.einsum("ijk,ia,jb,kc->abc", C, D[l1], D[l2], D[l3])
        assert torch.allclose(C, C2, atol=1e-9), (l1, l2, l3)


def test_normalize2mom_constants(golden):
    want = {"silu": 1.679176792399, "sigmoid": 1.846705534215, "tanh": 1.593733447259, "abs": 1.001110600838}
    fns = {"silu": torch.nn.functional.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "abs": torch.abs}
    got = [enn.normalize2mom(fns[k]).cst for k in ("silu", "sigmoid", "tanh", "abs")]
    for g, k in zip(got, ("silu", "sigmoid", "tanh", "abs")):
        assert abs(g - want[k]) < 1e-9
    assert np.allclose(got, golden["normalize2mom"], atol=1e-15)


def test_bessel_embedding_and_scatter():
    r = torch.tensor([0.0, 0.5, 2.5, 4.999, 5.0, 6.0])
    e = soft_one_hot_linspace(r, 0.0, 5.0, 8, basis="bessel", cutoff=True)
    assert e.shape == (6, 8)
    assert torch.all(e[4:] == 0)  # hard cutoff: x/c < 1 is strict
    k = torch.arange(1, 9)
    want = math.sqrt(2 / 5.0) * torch.sin(k * math.pi * 2.5 / 5.0) / 2.5
    assert torch.allclose(e[2], want, atol=1e-6)
    src = torch.arange(12.0).reshape(6, 2)
    idx = torch.tensor([0, 2, 2, 0, 5, 2])
    out = scatter(src, idx, dim_size=7)
    assert out.shape == (7, 2) and torch.equal(out[2], src[1] + src[2] + src[5]) and torch.all(out[1] == 0)
    mean = scatter(src, idx, reduce="mean")
    assert mean.shape == (6, 2) and torch.allclose(mean[2], (src[1] + src[2] + src[5]) / 3) and torch.all(mean[3] == 0)


def test_cartesian_tensor_basis_properties(golden):
    ct = io.CartesianTensor("ijkl=jikl=klij")
    assert str(ct) == "2x0e+2x2e+1x4e" and ct.dim == 21
    Q = ct.change_of_basis(torch.float64)
    Qf = Q.flatten(1)
    assert torch.allclose(Qf @ Qf.T, torch.eye(21, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(Q, Q.transpose(1, 2)) and torch.allclose(Q, Q.transpose(3, 4))
    assert torch.allclose(Q, Q.permute(0, 3, 4, 1, 2))
    g = torch.Generator().manual_seed(5)
    R = o3.rand_matrix(g)
    x = torch.randn(4, 21, dtype=torch.float64, generator=g)
    T = ct.to_cartesian(x)
    TR = torch.einsum("im,jn,kp,lq,bmnpq->bijkl", R, R, R, R, T)
    D = torch.block_diag(*[o3.wigner_D_from_sh(l, R) for l in [0, 0, 2, 2, 4]])
    assert torch.allclose(ct.to_cartesian(x @ D.T), TR, atol=1e-9)
    assert torch.allclose(ct.from_cartesian(T), x, atol=1e-12)
    assert np.allclose(Q.numpy(), golden["cart_basis_ijkl"], atol=1e-13)
    assert str(io.CartesianTensor("ij=ji")) == "1x0e+1x2e"


def test_paper_config_census():
    """paths / W / D_in / D_mid / FCTP parameter counts at S=86."""
    m = ScalarTensorOracle(dict(PAPER, average_num_neighbors=18.0), {"allowed_species": list(range(1, 87))})
    want = {
        "layer0_convnet.conv": (5, 80, 16, 400, 77056, 22016, 110080),
        "layer1_convnet.conv": (59, 452, 132, 2324, 238736, 112144, 586176),
        "layer2_convnet.conv": (99, 714, 214, 3658, 262472, 135880, 780536),
        "conv_layer_last": (103, 842, 246, 4170, 223944, 223944, 707608),
    }
    for name, w in want.items():
        c = m.backbone.get_submodule(name)
        tp = c.tp.tp
        got = (len(tp.instructions), tp.weight_numel, tp.irreps_in1.dim, tp.irreps_out.dim, c.sc.weight.numel(),
               c.lin1.weight.numel(), c.lin2.weight.numel())
        assert got == w, (name, got)
    assert m.backbone.conv_to_output_hidden.linear.weight.numel() == 522
    assert m.extra_layers_dict["out_layer"].weight.numel() == 37


def test_neighbor_list_contract(golden_dir, golden):
    a = 5.46
    lat = np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]])
    pos = np.array([[0.0, 0.0, 0.0], [0.25, 0.25, 0.25]]) @ lat
    ei, sh = rdata.neighbor_list(pos, lat, 5.0)
    assert ei.shape == (2, 56) and np.all(np.bincount(ei[0]) == 28)
    # symmetric: (i,j,S) present <=> (j,i,-S) present; no true self edge
    fwd = {(int(i), int(j), *map(int, s)) for i, j, s in zip(ei[0], ei[1], sh)}
    assert all((j, i, -sx, -sy, -sz) in fwd for (i, j, sx, sy, sz) in fwd)
    assert not any(i == j and (sx, sy, sz) == (0, 0, 0) for (i, j, sx, sy, sz) in fwd)
    d = pos[ei[1]] + sh @ lat - pos[ei[0]]
    assert np.all(np.linalg.norm(d, axis=1) < 5.0)
    structs = rdata.structures_from_json(os.path.join(golden_dir, "example_crystal_elasticity_tensor_n100.json"))
    counts = [rdata.neighbor_list(s["cart_coords"], s["lattice"], 5.0)[0].shape[1] for s in structs]
    assert sum(len(s["cart_coords"]) for s in structs) == 473 and sum(counts) == 14380
    assert min(counts) == 12 and max(counts) == 2624
    assert np.array_equal(np.array(counts), golden["n100_edges_per_crystal"])


def test_oracle_golden_outputs(golden_dir, golden):
    torch.set_num_threads(4)
    s = rdata.structures_from_json(os.path.join(golden_dir, "elastic_tensor_one.json"))[0]
    g = rdata.crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0)
    assert np.array_equal(g["edge_index"].numpy(), golden["teo_edge_index"])
    torch.manual_seed(35)
    m = ScalarTensorOracle(dict(EQUIV_TEST), {"allowed_species": [8, 52]}).eval()
    with torch.no_grad():
        got = m.decode(rdata.collate([g])).numpy()
    assert np.allclose(got, golden["teo_cartesian"], rtol=0, atol=2e-6 * np.abs(golden["teo_cartesian"]).max())

    structs = rdata.structures_from_json(os.path.join(golden_dir, "example_crystal_elasticity_tensor_n100.json"))[:6]
    graphs = [rdata.crystal_graph(t["cart_coords"], t["lattice"], t["atomic_numbers"], 5.0) for t in structs]
    species = [int(z) for z in golden["n100_species"]]
    assert len(species) == 73 and abs(float(golden["n100_avg_num_neigh"]) - 30.4017) < 1e-3
    torch.manual_seed(35)
    m = ScalarTensorOracle(dict(PAPER), {"allowed_species": species,
                                         "average_num_neighbors": float(golden["n100_avg_num_neigh"])}).eval()
    with torch.no_grad():
        got = m.decode(rdata.collate(graphs)).numpy()
    assert np.allclose(got, golden["n100_first6_irreps"], rtol=0, atol=2e-6 * np.abs(golden["n100_first6_irreps"]).max())


def test_lmax2_head_zero_fills_unreachable_4e():
    """with l<=2 features the 4e output of the head has no path and is zero."""
    from common import LMAX2
    from matten_amd.data import synthetic  # host-side generator only (no compute)

    graphs = synthetic.fcc64_graphs(1)
    torch.manual_seed(0)
    m = ScalarTensorOracle(dict(LMAX2), {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}).eval()
    with torch.no_grad():
        y = m.decode(rdata.collate(graphs))
    assert y.shape == (1, 21) and torch.all(y[:, 12:] == 0) and y[:, :12].abs().max() > 0


def test_atomic_tensor_oracle_is_equivariant_per_atom(golden_dir):
    """AtomicTensorModel (reference model_factory/tfn_atomic_tensor.py, config scripts/configs/atomic_tensor.yaml):
    one symmetric 2-tensor per atom, irreps 0e+2e; rotating the crystal rotates every atom's tensor
    (the property the reference checks for the crystal-level model, tests/model/test_tfn_tensor.py:98-139)."""
    from common import ATOMIC
    from oracle.e3nn_lite import o3
    from oracle.matten_ref.model import AtomicTensorOracle, ToCartesian

    s = rdata.structures_from_json(os.path.join(golden_dir, "elastic_tensor_one.json"))[0]
    torch.manual_seed(35)
    Q = o3.rand_matrix()
    hp = dict(ATOMIC, normalization=None, output_format="cartesian")
    ds = {"allowed_species": sorted(set(int(z) for z in s["atomic_numbers"])), "average_num_neighbors": 30.0}
    m = AtomicTensorOracle(hp, ds).eval()
    g1 = rdata.crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0)
    g2 = rdata.crystal_graph(s["cart_coords"] @ Q.numpy().T, s["lattice"] @ Q.numpy().T, s["atomic_numbers"], 5.0)
    with torch.no_grad():
        t1 = m(rdata.collate([g1]))
        t2 = m(rdata.collate([g2]))
    n = len(s["atomic_numbers"])
    assert t1.shape == (n, 3, 3) and t1.abs().max() > 1e-3
    assert torch.allclose(t1, t1.transpose(1, 2), atol=1e-6)
    want = torch.einsum("ia,jb,nab->nij", Q.to(t1.dtype), Q.to(t1.dtype), t1)
    assert torch.allclose(t2, want, atol=1e-4), (t2 - want).abs().max()
    # irreps output: 0e + 2e = 6 numbers per atom, and ToCartesian maps them onto the same tensors
    m_ir = AtomicTensorOracle(dict(hp, output_format="irreps"), ds).eval()
    m_ir.load_state_dict(m.state_dict())
    with torch.no_grad():
        ir = m_ir(rdata.collate([g1]))
    assert ir.shape == (n, 6)
    assert torch.allclose(ToCartesian("ij=ji")(ir), t1, atol=1e-6)


def test_su2_clebsch_gordan_matches_sympy():
    """Independent pin of the published part of the coupling coefficients: the oracle's su(2) Clebsch-Gordan
    coefficients <l1 m1 l2 m2 | l3 m3> (Racah formula, e3nn_lite/o3.py) against sympy's exact evaluation, every
    (l1, l2, l3) with l <= 4.  (The real-basis change on top of them is e3nn's convention)"""
    from sympy import S
    from sympy.physics.wigner import clebsch_gordan

    n = 0
    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1):
                C = o3._su2_clebsch_gordan(l1, l2, l3)
                for m1 in range(-l1, l1 + 1):
                    for m2 in range(-l2, l2 + 1):
                        m3 = m1 + m2
                        if abs(m3) > l3:
                            continue
                        want = float(clebsch_gordan(S(l1), S(l2), S(l3), S(m1), S(m2), S(m3)))
                        assert abs(C[l1 + m1, l2 + m2, l3 + m3].item() - want) < 1e-12, (l1, l2, l3, m1, m2)
                        n += 1
    assert n == 1439  # all (m1, m2) with |m1 + m2| <= l3 over the 65 triples


def test_real_spherical_harmonics_match_sympy():
    """The oracle's real harmonics against sympy's complex Ynm: with the polar axis on y, (x_s, y_s, z_s) = (z, x, y),
    and without the Condon-Shortley sign, norm-normalised:
        Y_{l,m>0} = sqrt(4 pi/(2l+1)) sqrt2 (-1)^m Re Ynm(l, m),  Y_{l,-m} = ... Im Ynm(l, m),  Y_{l,0} = ... Ynm(l, 0)."""
    import sympy as sp

    th, ph = sp.symbols("theta phi", real=True)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(6, 3, dtype=torch.float64, generator=g)
    n = v / v.norm(dim=1, keepdim=True)
    x, y, z = n[:, 0].numpy(), n[:, 1].numpy(), n[:, 2].numpy()
    xs, ys, zs = z, x, y
    theta, phi = np.arccos(zs), np.arctan2(ys, xs)
    got = o3.spherical_harmonics([0, 1, 2, 3, 4], v, True, "norm").numpy()
    for l in range(5):
        scale = math.sqrt(4 * math.pi / (2 * l + 1))
        for m in range(0, l + 1):
            f = sp.lambdify((th, ph), sp.Ynm(l, m, th, ph).expand(func=True), "numpy")
            val = np.asarray(f(theta, phi), dtype=np.complex128) * np.ones_like(theta)
            if m == 0:
                assert np.allclose(got[:, l * l + l], scale * val.real, atol=1e-12), (l, m)
            else:
                c = scale * math.sqrt(2) * (-1) ** m
                assert np.allclose(got[:, l * l + l + m], c * val.real, atol=1e-12), (l, m)
                assert np.allclose(got[:, l * l + l - m], c * val.imag, atol=1e-12), (l, -m)


def test_real_w3j_matches_gaunt_integrals():
    """Ties the one e3nn-specific step that sympy has not pinned -- the real-basis change on top of the su(2)
    coefficients -- to the harmonics that ARE pinned: for every triple (l1, l2, l3), l <= 4, with l1 + l2 + l3 even, the
    oracle's real ``wigner_3j`` must be proportional, entry for entry, to the Gaunt integral of its own real harmonics
        G_ijk = (1 / 4 pi) Int Y_{l1,i} Y_{l2,j} Y_{l3,k} dOmega
    with ONE positive factor per triple (the invariant 3-tensor of three irreps is unique up to scale; the sign is the
    convention SURVEY appendix A.2 records: c > 0 for every even-sum triple).  Product quadrature, exact for the
    polynomial degree <= 12 involved: 16-point Gauss-Legendre in cos(theta) x 32 uniform azimuths, fp64.  Odd-sum triples
    have a vanishing Gaunt integral (parity); they stay on the rotation-equivariance tests."""
    nodes, weights = np.polynomial.legendre.leggauss(16)
    nphi = 32
    phi = 2 * np.pi * np.arange(nphi) / nphi
    ct, ph = np.meshgrid(nodes, phi, indexing="ij")
    st = np.sqrt(1 - ct**2)
    # any orthonormal frame does: the integral is over the whole sphere
    pts = torch.as_tensor(np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3))
    w = torch.as_tensor((weights[:, None] * np.full((1, nphi), 1.0 / nphi) / 2.0).reshape(-1))   # sums to 1 = dOmega / 4 pi
    Y = o3.spherical_harmonics([0, 1, 2, 3, 4], pts, True, "norm")
    assert abs(w.sum().item() - 1.0) < 1e-14
    # the quadrature reproduces the harmonics' orthogonality: (1/4pi) Int Y_a Y_b = delta_ab / (2l+1) in 'norm'
    gram = torch.einsum("p,pa,pb->ab", w, Y, Y)
    want = torch.diag(torch.cat([torch.full((2 * l + 1,), 1.0 / (2 * l + 1), dtype=torch.float64) for l in range(5)]))
    assert torch.allclose(gram, want, atol=1e-13)
    n = 0
    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1):
                G = torch.einsum("p,pi,pj,pk->ijk", w, Y[:, l1 * l1:(l1 + 1) ** 2], Y[:, l2 * l2:(l2 + 1) ** 2],
                                 Y[:, l3 * l3:(l3 + 1) ** 2])
                if (l1 + l2 + l3) % 2:
                    assert G.abs().max().item() < 1e-13, (l1, l2, l3)
                    continue
                C = o3.wigner_3j(l1, l2, l3, dtype=torch.float64)
                c = (C * G).sum() / (G * G).sum()
                assert c.item() > 0, (l1, l2, l3, c.item())
                assert torch.allclose(C, c * G, atol=1e-12), (l1, l2, l3, (C - c * G).abs().max().item())
                n += 1
    assert n == 42   # even-sum triples with l <= 4 (of 65)


def test_oracle_matches_e3nn_golden(golden_dir):
    """tests/golden/e3nn_golden.npz is written by tests/golden/make_golden_e3nn.py where REAL e3nn 0.5.1 and the
    reference run (neither exists in the build container nor on the GPU box).  When the file is there, every e3nn-held
    piece of the oracle is compared with it: Wigner-3j for all l <= 4 triples, spherical harmonics, the Cartesian bases,
    the normalize2mom constants, the Bessel embedding and -- with the reference model's own state_dict loaded into the
    oracle -- the node features after every backbone module and the Cartesian tensor of the TeO fixture
    (tests/model/test_tfn_tensor.py:23-42,99 of the reference).  Without the file the oracle is pinned only by the
    reference's property tests, the survey's anchors and sympy: PARITY UNPINNED, reported as a skip."""
    path = os.path.join(golden_dir, "e3nn_golden.npz")
    if not os.path.exists(path):
        pytest.skip("PARITY UNPINNED: tests/golden/e3nn_golden.npz absent (e3nn is not installable here; "
                    "generate it with tests/golden/make_golden_e3nn.py where e3nn==0.5.1 runs)")
    ref = np.load(path)
    for key in ref.files:
        if key.startswith("w3j_"):
            l1, l2, l3 = map(int, key.split("_")[1:])
            got = o3.wigner_3j(l1, l2, l3, dtype=torch.float64).numpy()
            assert np.allclose(got, ref[key], atol=1e-12), key
    pts = torch.as_tensor(ref["sh_points"])
    assert np.allclose(o3.spherical_harmonics([0, 1, 2, 3, 4], pts, True, "component").numpy(), ref["sh_values"], atol=1e-12)
    for name, formula in (("ijkl", "ijkl=jikl=klij"), ("ij", "ij=ji")):
        ct = io.CartesianTensor(formula)
        assert str(ct) == str(ref[f"cart_irreps_{name}"])
        assert np.allclose(ct.change_of_basis(torch.float64).numpy(), ref[f"cart_basis_{name}"], atol=1e-10), formula
    acts = {"silu": torch.nn.functional.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "abs": torch.abs,
            "ssp": lambda x: torch.nn.functional.softplus(x) - math.log(2.0)}
    for k, f in acts.items():
        assert abs(enn.normalize2mom(f).cst - float(ref[f"normalize2mom_{k}"])) < 1e-6, k
    r = torch.as_tensor(ref["soft_one_hot_r"])
    assert np.allclose(soft_one_hot_linspace(r, 0.0, 5.0, 8, "bessel", True).numpy(), ref["soft_one_hot_bessel"], atol=1e-12)

    # the reference's model on the TeO fixture, layer by layer
    batch = {k[len("teo/in/"):]: torch.as_tensor(ref[k]) for k in ref.files if k.startswith("teo/in/")}
    state = {k[len("teo/state/"):]: torch.as_tensor(ref[k]) for k in ref.files if k.startswith("teo/state/")}
    model = create_model(dict(EQUIV_TEST), {"allowed_species": [8, 52]}).eval()
    own = model.state_dict()
    missing = [k for k in own if k not in state]
    assert not missing, missing
    model.load_state_dict({k: state[k] for k in own})
    acts_out = {}
    for name, child in model.named_children():
        child.register_forward_hook(
            lambda mod, inp, res, name=name: acts_out.__setitem__(name, res["node_features"].clone())
            if isinstance(res, dict) and "node_features" in res else None)
    with torch.no_grad():
        res = model(dict(batch))
    for k in ref.files:
        if k.startswith("teo/act/"):
            want = torch.as_tensor(ref[k])
            got = acts_out[k[len("teo/act/"):]]
            assert torch.allclose(got, want, rtol=0, atol=2e-5 * want.abs().max().item()), k
    want = torch.as_tensor(ref["teo/my_model_output"])
    assert torch.allclose(res["my_model_output"], want, rtol=0, atol=2e-5 * want.abs().max().item())
    cart = ToCartesian("ijkl=jikl=klij")(res["my_model_output"])
    want = torch.as_tensor(ref["teo/cartesian"])
    assert torch.allclose(cart, want, rtol=0, atol=2e-5 * want.abs().max().item())
