"""
CPU tests of the host side: irreps algebra and planners, the C-ABI library (loads and exports every
symbol the header declares -- no kernel is launched), graph construction, batch sharding over a
2-rank gloo group, and API/state_dict compatibility with the reference's names.
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import EQUIV_TEST, LMAX2, PAPER

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from matten_amd import _lib

    header = open(os.path.join(ROOT, "include", "matten_hip.h")).read()
    declared = set(re.findall(r"\b(matten_[a-z0-9_]+)\s*\(", header)) - {"matten_stream_t"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.matten_abi_version() == _lib.ABI_VERSION
    from matten_amd.plan import TP_TILE_NODES

    assert lib.matten_tp_tile_nodes() == TP_TILE_NODES
    # the measurement helpers live in their own library with their own header (include/matten_lab.h): none of them in the
    # product, all of them exported there
    from matten_amd import lab

    lab_header = open(os.path.join(ROOT, "include", "matten_lab.h")).read()
    lab_declared = set(re.findall(r"\b(matten_[a-z0-9_]+)\s*\(", lab_header)) - {"matten_lab_stream_t"}
    assert lab_declared == set(lab.SIGNATURES) and not (lab_declared & declared)
    assert not any(hasattr(lib, n) for n in lab_declared)
    lablib = lab.load()
    assert lablib is not None and all(hasattr(lablib, n) for n in lab_declared)
    # host-detectable argument errors are reported without touching a GPU
    assert lib.matten_radial_mlp(None, -1, 8, 0.0, 5.0, None, 8, None, None, 32, 848, 1.0, None, 0, None) == -1
    assert lib.matten_species_linear(None, 0, None, None, 1, None, 0, None, 0, 0, None, 0, 1, None, None) == -1


def test_generated_cg_header_is_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "matten_amd", "csrc", "gen_cg.py")],
                         capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "matten_amd", "csrc", "cg_gen.h")).read()


def test_product_o3_matches_oracle_tables():
    from matten_amd import o3 as p
    from oracle.e3nn_lite import io, o3 as r

    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), l1 + l2 + 1):
                assert np.allclose(p.wigner_3j(l1, l2, l3), r.wigner_3j(l1, l2, l3, dtype=torch.float64).numpy(),
                                   atol=1e-13)
    for f in ("ijkl=jikl=klij", "ij=ji"):
        irreps, Q = p.cartesian_tensor_basis(f)
        ct = io.CartesianTensor(f)
        assert str(irreps) == str(ct)
        assert np.allclose(Q, ct.change_of_basis(torch.float64).numpy(), atol=1e-13)
    a = p.Irreps("2x1o+3x0e+1x0o+4x0e")
    b = r.Irreps("2x1o+3x0e+1x0o+4x0e")
    assert str(a.sort()[0]) == str(b.sort()[0]) and a.sort()[1] == b.sort()[1]
    assert str(a.sort()[0].simplify()) == str(b.sort()[0].simplify())


@pytest.mark.parametrize("hp,S", [(PAPER, 86), (EQUIV_TEST, 2), (LMAX2, 10)])
def test_plans_mirror_the_oracle_instruction_lists(hp, S):
    """Same paths, slots, weight layout and irreps as the oracle's e3nn-style modules, layer by layer."""
    from matten_amd.model_factory.tfn_scalar_tensor import create_model as create_product
    from oracle.matten_ref.model import create_model as create_oracle

    ds = {"allowed_species": list(range(1, S + 1)), "average_num_neighbors": 18.0}
    ref, prod = create_oracle(dict(hp), ds), create_product(dict(hp), ds)
    names = [n for n, _ in ref.named_children()]
    assert names == [n for n, _ in prod.named_children()]
    for name in names:
        r, p = ref.get_submodule(name), prod.get_submodule(name)
        for key in r.irreps_out:
            assert str(r.irreps_out[key]) == str(p.irreps_out[key]), (name, key)
        conv_r = getattr(r, "conv", r if name == "conv_layer_last" else None)
        if conv_r is None:
            continue
        conv_p = getattr(p, "conv", p)
        tp_r, plan = conv_r.tp.tp, conv_p.tp.plan
        assert len(tp_r.instructions) == len(plan.paths) and tp_r.weight_numel == plan.weight_numel
        assert str(tp_r.irreps_out) == str(plan.irreps_mid)
        out_offs = [s.start for s in tp_r.irreps_out.slices()]
        in_offs = [s.start for s in tp_r.irreps_in1.slices()]
        w = 0
        for ins, path in zip(tp_r.instructions, plan.paths):
            assert (ins.i_in1, ins.i_in2) == (path.i_in1, path.i_sh)
            assert out_offs[ins.i_out] == path.out_off and in_offs[ins.i_in1] == path.x_off and path.w_off == w
            assert abs(ins.path_weight - (2 * path.l3 + 1) ** 0.5) < 1e-12
            w += path.mul
        for lin in ("sc", "lin1", "lin2"):
            assert getattr(conv_r, lin).weight.numel() == getattr(conv_p, lin).plan.weight_numel
            assert str(getattr(conv_r, lin).irreps_out) == str(getattr(conv_p, lin).plan.irreps_out)
        if hasattr(r, "act"):
            assert str(r.act.irreps_in) == str(p.act.irreps_in) and str(r.act.irreps_out) == str(p.act.irreps_out)


def test_state_dict_names_match_reference_contract():
    """SURVEY.md Appendix C parameter names; oracle <-> product state_dicts interchange."""
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
    from oracle.matten_ref.model import ScalarTensorOracle

    ds = {"allowed_species": [13, 29, 79], "average_num_neighbors": 18.0}
    ref = ScalarTensorOracle(dict(PAPER), ds)
    model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds)
    sd = model.state_dict()
    for k in ("backbone.one_hot.linear.weight", "backbone.one_hot.atomic_number_to_index._Z_to_index",
              "backbone.layer0_convnet.conv.sc.weight", "backbone.layer1_convnet.conv.lin1.weight",
              "backbone.layer2_convnet.conv.lin2.weight", "backbone.layer0_convnet.conv.tp.weight_nn.layer2.weight",
              "backbone.layer2_convnet.norm.n.running_var", "backbone.conv_layer_last.tp.weight_nn.layer0.weight",
              "backbone.conv_to_output_hidden.linear.weight", "extra_layers_dict.out_layer.weight"):
        assert k in sd, k
    assert sd["backbone.conv_to_output_hidden.linear.weight"].numel() == 522
    assert sd["extra_layers_dict.out_layer.weight"].numel() == 37
    missing, unexpected = model.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and all(k.endswith(("output_mask", "tp.tp.weight")) for k in unexpected)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items() if k in sd}
    assert model.hparams["dataset_hparams"]["allowed_species"] == [13, 29, 79]  # reference predict.py:100
    assert model.to_cartesian is None and hasattr(model, "backbone") and "out_layer" in model.extra_layers_dict


def test_no_cpu_fallback_and_loud_errors():
    from matten_amd import _lib
    from matten_amd.data import synthetic
    from matten_amd.data.graph import collate
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).eval()
    with pytest.raises(_lib.MattenHipError, match="no CPU fallback"):
        model(collate(synthetic.fcc64_graphs(1)))
    import glob

    for path in glob.glob(os.path.join(ROOT, "matten_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path
    with pytest.raises(NotImplementedError, match="radial_basis_type"):  # the up-front validator
        ScalarTensorModel(backbone_hparams=dict(PAPER, radial_basis_type="gaussian"), dataset_hparams=ds)
    # a module built directly still refuses what it does not implement, wrapped like the reference's factory does
    from matten_amd.model_factory.utils import create_sequential_module
    from matten_amd.nn.embedding import EdgeLengthEmbedding, SpeciesEmbedding
    with pytest.raises(RuntimeError, match="EdgeLengthEmbedding") as ei:
        create_sequential_module({"one_hot": (SpeciesEmbedding, {"allowed_species": [13]}),
                                  "radial_basis": (EdgeLengthEmbedding, {"basis": "gaussian"})})
    assert isinstance(ei.value.__cause__, NotImplementedError)
    with pytest.raises(RuntimeError, match="Failed instantiate module"):
        ScalarTensorModel(backbone_hparams=dict(PAPER, conv_layer_irreps="4x5e"), dataset_hparams=ds)


def test_graph_builder_matches_oracle_bit_exact(golden_dir):
    from matten_amd.data import graph, synthetic
    from oracle.matten_ref import data as rdata

    structs = rdata.structures_from_json(os.path.join(golden_dir, "example_crystal_elasticity_tensor_n100.json"))
    structs += synthetic.fcc64_structures(3)
    for s in structs:
        a, b = graph.neighbor_list(s["cart_coords"], s["lattice"], 5.0)
        c, d = rdata.neighbor_list(s["cart_coords"], s["lattice"], 5.0)
        assert np.array_equal(a, c) and np.array_equal(b, d)
    gs = synthetic.fcc64_graphs(4)
    assert all(g["edge_index"].shape == (2, 1152) and torch.all(g["num_neigh"] == 18) for g in gs)
    b = graph.collate(gs)
    want = rdata.collate([rdata.crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0)
                          for s in synthetic.fcc64_structures(4)])
    assert set(b) == set(want)
    for k in want:
        assert torch.equal(b[k], want[k]) and b[k].dtype == want[k].dtype, k
    assert b["cell"].shape == (12, 3) and b["ptr"].tolist() == [0, 64, 128, 192, 256]
    with pytest.raises(ValueError, match="no edges remain"):
        graph.neighbor_list(np.zeros((1, 3)), 50.0 * np.eye(3), 5.0)


def test_shard_bounds_cover_and_balance():
    from matten_amd.parallel import shard_bounds

    for n in (0, 1, 7, 8, 1000, 8001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _gloo_worker(rank, world, port, n_items, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matten_amd.parallel import shard_bounds, sharded_apply

    items = list(range(n_items))
    calls = []

    def fn(shard):  # stand-in for the per-rank forward: a deterministic function of the crystal index
        calls.append(list(shard))
        idx = torch.tensor(shard, dtype=torch.float32)
        return torch.stack([idx * (k + 1) for k in range(21)], dim=1)

    out = sharded_apply(fn, items, (21,), "cpu")
    lo, hi = shard_bounds(n_items, rank, world)
    assert calls == ([items[lo:hi]] if hi > lo else [])
    torch.save(out, os.path.join(tmp, f"out{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_sharding_gathers_in_batch_order(tmp_path, n_items):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() + n_items) % 2000
    mp.spawn(_gloo_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    want = torch.stack([torch.arange(n_items, dtype=torch.float32) * (k + 1) for k in range(21)], dim=1)
    for r in range(2):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"out{r}.pt")), want)


def test_predict_coalesces_user_batches_within_the_node_budget():
    """evaluate_soa merges consecutive user batches (reference predict.py:155 batch_size) while the merged forward stays
    within the atom budget: every crystal exactly once, in order; a batch above the budget stays whole; batch_size still
    caps a forward when the crystals are large."""
    from matten_amd.predict import coalesce_batches

    sizes = np.array([5] * 1000)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    ch = coalesce_batches(ptr, 200, node_budget=65536)
    assert len(ch) == 1 and (ch[0] == np.arange(1000)).all()
    ch = coalesce_batches(ptr, 200, node_budget=2400)          # 480 crystals' worth: two user batches fit, three do not
    assert [len(c) for c in ch] == [400, 400, 200] and (np.concatenate(ch) == np.arange(1000)).all()
    big = np.concatenate([[0], np.cumsum(np.array([64] * 1000))])
    ch = coalesce_batches(big, 200, node_budget=10000)          # a user batch is 12800 atoms: above the budget, kept whole
    assert [len(c) for c in ch] == [200] * 5
    from matten_amd.predict import NODE_BUDGET, effective_node_budget

    # batch_size is the reference's memory knob (predict.py:155): lowered below its default of 200 nothing is merged; an
    # explicit node_budget wins either way (advisor finding of round 4)
    assert effective_node_budget(200) == NODE_BUDGET and effective_node_budget(1000) == NODE_BUDGET
    assert effective_node_budget(50) == 0 and effective_node_budget(50, 4096) == 4096 and effective_node_budget(500, 0) == 0
    ch = coalesce_batches(ptr, 50, node_budget=effective_node_budget(50))
    assert all(len(c) <= 50 for c in ch) and np.array_equal(np.concatenate(ch), np.arange(len(ptr) - 1))
    ch = coalesce_batches(big, 7, node_budget=64 * 20)          # ragged tail
    assert (np.concatenate(ch) == np.arange(1000)).all() and max(len(c) for c in ch) <= 21


def test_pack_structures_fast_path_equals_the_careful_one(monkeypatch):
    """predict.pack_structures: well-formed dicts of numpy arrays take a vectorised path (no per-structure Python body);
    anything else -- malformed, non-finite, singular, pymatgen-like objects -- goes through the per-structure path that
    warns and skips like the reference's dataset loop (dataset/structure_scalar_tensor.py:296-362).  Same arrays either way."""
    from matten_amd import predict as P
    from matten_amd.data import synthetic

    st = synthetic.fcc64_structures(7)
    fast = P._pack_fast(st)
    assert fast is not None
    monkeypatch.setattr(P, "_pack_fast", lambda s: None)
    slow = P.pack_structures(st)
    monkeypatch.undo()
    assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(fast[:4], slow[:4])) and fast[4:] == slow[4:]
    malformed = {"lattice": np.eye(3), "cart_coords": np.zeros((2, 3)), "atomic_numbers": np.array([29])}
    singular = dict(st[0], lattice=np.zeros((3, 3)))
    nonfinite = dict(st[0], cart_coords=np.full((64, 3), np.nan))
    lists = dict(st[0], cart_coords=st[0]["cart_coords"].tolist())
    for bad in (malformed, singular, nonfinite, lists):
        assert P._pack_fast(st[:2] + [bad]) is None
    with pytest.warns(UserWarning, match="structure 2"):
        pos, cell, Z, ptr, keep, failed = P.pack_structures(st[:2] + [malformed] + st[2:4])
    assert keep == [0, 1, 3, 4] and failed == [2] and len(ptr) == 5
    pos2, _, _, ptr2, keep2, failed2 = P.pack_structures(st[:2] + [lists])     # lists are fine for the careful path
    assert keep2 == [0, 1, 2] and not failed2 and ptr2[-1] == 192


def test_predict_api_surface():
    import inspect

    from matten_amd import predict as P

    sig = inspect.signature(P.predict)
    ref_params = ["structure", "model_identifier", "checkpoint", "batch_size", "logger_level", "is_elasticity_tensor",
                  "is_atomic_tensor"]  # reference predict.py:151-159
    assert list(sig.parameters)[: len(ref_params)] == ref_params
    assert sig.parameters["model_identifier"].default == "20230627" and sig.parameters["batch_size"].default == 200
    with pytest.raises(FileNotFoundError, match="model_final.ckpt"):
        P.get_pretrained_model("20230627")

    class M:
        hparams = {"dataset_hparams": {"allowed_species": [14]}}

    with pytest.raises(RuntimeError, match="not supported by the model"):
        P.check_species(M(), [{"lattice": np.eye(3), "cart_coords": np.zeros((1, 3)), "atomic_numbers": [8]}])
    graphs, failed = P.build_graphs(
        [{"lattice": 3.0 * np.eye(3), "cart_coords": np.zeros((1, 3)), "atomic_numbers": [14]},
         {"lattice": 50.0 * np.eye(3), "cart_coords": np.zeros((1, 3)), "atomic_numbers": [14]}], 5.0)
    assert len(graphs) == 1 and failed == [1]


def test_nodewise_select_known_answer():
    """reference tests/nn/test_nodewise.py:7-30, verbatim scenario"""
    from matten_amd.data.irreps import DataKey
    from matten_amd.nn.nodewise import NodewiseSelect

    node_feats, mask_field, out_field = DataKey.NODE_FEATURES, "node_masks", "selected_node_features"
    aws = NodewiseSelect(irreps_in={node_feats: None, mask_field: None}, field=node_feats, out_field=out_field,
                         mask_field=mask_field)
    n_atoms = 5
    data = {node_feats: torch.arange(n_atoms * 2).reshape(n_atoms, 2),
            mask_field: torch.tensor([True, False, True, True, False])}
    out = aws(data)
    assert torch.allclose(out[out_field], data[node_feats][data[mask_field]])
    assert out_field not in data  # the input dict is left alone
    assert torch.equal(NodewiseSelect(irreps_in={node_feats: None}, field=node_feats)(data)[node_feats], data[node_feats])


def test_target_normalizers_reference_scenarios():
    """reference tests/data/test_transform.py:7-72 (MeanNormNormalize on 0e+2x1e+2e, ScalarNormalize), same checks;
    and the model-side use: ScalarTensorModel.transform_prediction applies task.normalizer.inverse."""
    from matten_amd.data.transform import MeanNormNormalize, ScalarNormalize

    N, dim = 3, 1 + 2 * 3 + 5
    data = torch.arange(N * dim).reshape(N, dim).to(torch.float)
    mnn = MeanNormNormalize("0e+2x1e+2e", eps=0.0)
    with pytest.raises(RuntimeError, match="not initialized"):
        mnn(data)
    mean1, norm1 = mnn.compute_statistics(data)
    sd = mnn.state_dict()
    mean, norm = sd["mean"], sd["norm"]
    assert torch.allclose(mean1, mean) and torch.allclose(norm1, norm)
    assert mean.shape == torch.Size((dim,)) and norm.shape == torch.Size((dim,))
    scalars = data[:, 0]
    assert mean[0] == scalars.mean() and norm[0] == torch.std(scalars, unbiased=False)
    n = data[:, 4:7].square().mean(dim=1).mean(dim=0).sqrt()  # the second 1e lives in columns 4..6
    assert torch.allclose(norm[4], n) and torch.allclose(norm[5], n) and torch.allclose(norm[6], n)
    assert mean[1:].abs().max() == 0  # only scalars are centred
    assert torch.allclose(data, mnn.inverse(mnn(data)))

    data = torch.arange(3 * 4).reshape(3, 4).to(torch.float)
    sn = ScalarNormalize(num_features=4)
    mean1, norm1 = sn.compute_statistics(data)
    sd = sn.state_dict()
    assert torch.allclose(mean1, sd["mean"]) and torch.allclose(norm1, sd["norm"])
    assert torch.allclose(sd["mean"], torch.mean(data, dim=0))
    assert torch.allclose(sd["norm"], torch.std(data, dim=0, unbiased=False))
    assert torch.allclose(data, sn.inverse(sn(data)))

    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    class Task:
        name = "elastic_tensor_full"
        normalizer = mnn

    x = torch.randn(2, dim)
    out = ScalarTensorModel.transform_prediction(type("M", (), {"tasks": {"elastic_tensor_full": Task()}})(),
                                                 {"elastic_tensor_full": x})
    assert torch.allclose(out["elastic_tensor_full"], mnn.inverse(x))


def test_dataset_json_reader_matches_oracle_reader(golden_dir):
    """matten_amd.data.io reads the reference's dataset format (dataset/structure_scalar_tensor.py:229-243) like the
    oracle's reader: 100 rows, 473 atoms, same arrays."""
    from matten_amd.data.io import structures_from_json
    from oracle.matten_ref import data as rdata

    path = os.path.join(golden_dir, "example_crystal_elasticity_tensor_n100.json")
    got, want = structures_from_json(path), rdata.structures_from_json(path)
    assert len(got) == len(want) == 100 and sum(len(s["atomic_numbers"]) for s in got) == 473
    for a, b in zip(got, want):
        for k in ("lattice", "cart_coords", "atomic_numbers", "elastic_tensor_full"):
            assert np.array_equal(a[k], b[k]), k


def test_fused_unit_map_covers_every_entry_and_node_group_once():
    """plan.fused_unit_map: every (entry, node group) appears exactly once as a working unit; shared workgroups
    (four consecutive units) are homogeneous in node group and lanes-per-node and padded with loader-only units;
    paired workgroups hold two entries on two consecutive node groups."""
    from matten_amd import plan as mp
    from matten_amd.o3 import Irreps

    cases = [(irr, spec) for irr in ("32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e", "16x0e", "3x0e+5x1o+1x2e",
                                     "64x0e+1x1o+7x2e+2x4e") for spec in ("", "16:4,8:2,4:2,2:2", "16:8,8:8,4:8,2:8")]
    for irr, spec in cases:
        os.environ["MATTEN_TP_PERSIST"] = spec
        p = mp.plan_uvu(irr, Irreps.spherical_harmonics(4), irr)
        os.environ.pop("MATTEN_TP_PERSIST")
        for order in ("node", "entry"):
            os.environ["MATTEN_TP_PERSIST"] = spec
            m = mp.fused_unit_map(p.group_entries, order)
            work = []
            for v in (int(v) for v in m if not int(v) & mp.FUSED_UNIT_LOADER_ONLY):
                reps, step = 1 << ((v >> mp.FUSED_UNIT_REPS_SHIFT) & 3), 2 if v & mp.FUSED_UNIT_PAIRED else 1
                work += [((v >> 8) & 0xFFFF, (v & 255) + step * k) for k in range(reps)]   # a persistent unit: its groups in turn
            want = set()
            for e, row in enumerate(p.group_entries):
                if int(row[0]) < 0:
                    continue   # second-half record of a merged entry (plan.TP_KIND_MERGED): carried by the unit before it
                npw = max(1, 64 >> int(row[3]))
                want |= {(e, r) for r in range(-(-mp.TP_TILE_NODES // npw))}
            assert len(work) == len(set(work)) and set(work) == want
            shared = [bool(int(v) & mp.FUSED_UNIT_SHARED) for v in m]
            assert order == "node" or not any(shared)
            n_shared = sum(shared)
            assert n_shared % 4 == 0 and all(shared[:n_shared]) and not any(shared[n_shared:])
            for b in range(0, n_shared, 4):
                blk = [int(v) for v in m[b:b + 4]]
                if blk[0] & mp.FUSED_UNIT_PAIRED:   # two entries x two consecutive node groups: waves 0,1 | 2,3
                    assert all(v & mp.FUSED_UNIT_PAIRED for v in blk)
                    r = blk[0] & 255
                    assert [v & 255 for v in blk] == [r, r, r + 1, r + 1] and r % 2 == 0
                    assert (blk[0] >> 8 & 0xFFFF) == (blk[2] >> 8 & 0xFFFF) and (blk[1] >> 8 & 0xFFFF) == (blk[3] >> 8 & 0xFFFF)
                    assert max(1, 64 >> int(p.group_entries[blk[0] >> 8 & 0xFFFF][3])) <= 16
                else:
                    assert len({v & 255 for v in blk}) == 1
                assert len({int(p.group_entries[v >> 8 & 0xFFFF][3]) for v in blk}) == 1
                assert int(p.group_entries[blk[0] >> 8 & 0xFFFF][3]) >= 1
                assert not blk[0] & mp.FUSED_UNIT_LOADER_ONLY
                assert len({(v >> mp.FUSED_UNIT_REPS_SHIFT) & 3 for v in blk}) == 1   # one barrier sequence per workgroup
    os.environ.pop("MATTEN_TP_PERSIST", None)


def test_split_a_tiles_reconstructs_the_last_radial_layer():
    """ops.split_a_tiles (pure torch, runs on the CPU too): every fragment element is hi + 2^-11 lo of the scaled weight
    to 2^-22 relative of the entry's largest magnitude, in the lane layout include/matten_hip.h documents."""
    import torch

    from matten_amd import ops, plan as mp
    from matten_amd.o3 import Irreps

    irr = "32x0o+32x0e+16x1o+16x1e+4x2o+4x2e+2x3o+2x3e+2x4e"
    p = mp.plan_uvu(irr, Irreps.spherical_harmonics(4), irr)
    w_pad = (len(p.fused_cols) + 15) // 16 * 16 + 16
    g = torch.Generator().manual_seed(3)
    w2p = torch.randn(32, w_pad, generator=g) * torch.logspace(-6, 3, w_pad)[None, :]  # wide dynamic range over columns
    frag, inv = ops.split_a_tiles(w2p, p.group_entries)
    assert frag.shape == (p.fused_a_tiles, 64, 16) and frag.dtype == torch.float16 and inv.shape == (len(p.group_entries),)
    assert torch.isfinite(frag.float()).all()
    for e, row in enumerate(p.group_entries):
        w_base, t0, n_mt = int(row[5]), int(row[6]), int(row[7])
        if int(row[0]) < 0:      # second-half record of a merged entry: no weight tiles of its own, scale 1
            assert n_mt == 0 and float(inv[e]) == 1.0
            continue
        assert n_mt == -(-(int(row[2]) * bin(int(row[4]) & 0xFFFFFFFF).count("1")) // 16) or n_mt >= 1
        block = w2p[:, w_base:w_base + 16 * n_mt]
        scale = 1.0 / inv[e].item()
        assert 2.0 ** 13 <= block.abs().max().item() * scale < 2.0 ** 14
        for mt in range(n_mt):
            for lane in (0, 17, 42, 63):
                gi, c = lane >> 4, lane & 15
                for kk in range(8):
                    k = 16 * (kk >> 2) + 4 * gi + (kk & 3)
                    want = w2p[k, w_base + 16 * mt + c].item()
                    got = (frag[t0 + mt, lane, kk].float().item() + frag[t0 + mt, lane, 8 + kk].float().item() / 2048.0) * inv[e].item()
                    assert abs(got - want) <= 2.0 ** -21 * block.abs().max().item() + 1e-30


def test_unsupported_configs_fail_up_front_with_the_full_list():
    """every hyper-parameter outside the accelerated envelope is reported by ONE error at construction"""
    from common import PAPER
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
    from matten_amd.model_factory.utils import UnsupportedConfig, validate_hparams

    ds = {"allowed_species": [13, 29], "average_num_neighbors": 18.0}
    validate_hparams(PAPER, ds)  # the paper config is inside
    bad = dict(PAPER, nonlinearity_type="relu", normalization="layer", reduce="median", radial_basis_type="gaussian",
               invariant_neurons=64, use_atom_feats=True, irreps_edge_sh="0e+1o+2e+3o+4e+5o")
    with pytest.raises(UnsupportedConfig) as ei:
        ScalarTensorModel(backbone_hparams=bad, dataset_hparams=ds)   # (use_atom_feats without atom_feats_size)
    msg = str(ei.value)
    for needle in ("nonlinearity_type", "normalization", "reduce", "radial_basis_type", "invariant_neurons",
                   "use_atom_feats", "irreps_edge_sh"):
        assert needle in msg, needle
    assert isinstance(ei.value, NotImplementedError)
    with pytest.raises(UnsupportedConfig, match="allowed_species"):
        validate_hparams(PAPER, {"average_num_neighbors": 18.0})


def test_debug_log_level_inserts_anomaly_detectors():
    """reference model_factory/utils.py:82-87: with the log level at DEBUG every layer is followed by a DetectAnomaly
    module (same names: 'DetectAnomaly_<layer>'); parameters and state_dict keys are unchanged"""
    from common import LMAX2
    from matten_amd import log
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
    from matten_amd.nn.utils import DetectAnomaly
    from matten_amd.utils import detect_nan_and_inf

    ds = {"allowed_species": [13, 29], "average_num_neighbors": 18.0}
    hp = dict(LMAX2, num_layers=1)
    plain = ScalarTensorModel(backbone_hparams=hp, dataset_hparams=ds)
    log.set_logger("DEBUG", stderr=False)
    try:
        dbg = ScalarTensorModel(backbone_hparams=hp, dataset_hparams=ds)
    finally:
        log.set_logger("ERROR", stderr=False)
    names = [n for n, _ in dbg.backbone.named_children()]
    assert names[:4] == ["one_hot", "DetectAnomaly_one_hot", "spharm_edges", "DetectAnomaly_spharm_edges"]
    assert sum(isinstance(m, DetectAnomaly) for m in dbg.backbone.children()) == len(list(plain.backbone.children()))
    assert list(dbg.state_dict().keys()) == list(plain.state_dict().keys())
    # the detector itself (host tensors): NaN / Inf raise with the key and the layer name, non-float entries are skipped
    det = DetectAnomaly(irreps_in=None, name="layer0_convnet")
    ok = {"pos": torch.zeros(3, 3), "edge_index": torch.zeros(2, 4, dtype=torch.int64), "none": None, "pair": (1, 2)}
    assert det(ok) is ok
    with pytest.raises(ValueError, match="Anomaly detected for node_features of layer0_convnet"):
        det(dict(ok, node_features=torch.tensor([1.0, float("nan")])))
    with pytest.raises(ValueError, match="inf"):
        detect_nan_and_inf(torch.tensor([float("inf")]), file="f", name="x")


def test_bench_launches_its_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` (no RANK in the environment) must start torch.distributed.run itself, as a CHILD
    process, before anything imports torch.cuda; under torch.distributed.run it must run as a rank instead."""
    import importlib
    import subprocess
    import sys

    import bench

    importlib.reload(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MATTEN_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench, "run_rank", lambda args: (_ for _ in ()).throw(AssertionError("parent must not run a rank")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one rank + --force-dist: still a launch (RCCL path with world size 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--force-dist"])
    with pytest.raises(SystemExit):
        bench.main()
    assert "--nproc-per-node=1" in seen["cmd"] and seen["env"]["MATTEN_FORCE_DIST"] == "1"
    # as a rank of torch.distributed.run: no second launch
    ran = {}
    monkeypatch.setattr(bench, "run_rank", lambda args: ran.setdefault("args", args))
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    seen.clear()
    bench.main()
    assert ran["args"].gpus == 8 and not seen


def test_bench_shards_are_contiguous_slices_of_one_set():
    """SURVEY.md 8d config 5 / DESIGN.md section 6: with N > 1 ranks, rank r owns crystals [r B, (r + 1) B) of ONE set of
    N B crystals (seed FCC_SEED + 1), not N differently seeded sets; one rank runs the config-3 set (seed FCC_SEED).
    Checked on the structures (the graphs are a deterministic function of them) and on bench.py's call."""
    import inspect

    import bench
    from matten_amd.data import synthetic as S

    whole = S.fcc64_structures(12, S.FCC_SEED + 1)
    for r in range(3):
        part = S.fcc64_structures(4, S.FCC_SEED + 1, start=4 * r)
        for a, b in zip(whole[4 * r: 4 * r + 4], part):
            assert all(np.array_equal(a[k], b[k]) for k in ("lattice", "cart_coords", "atomic_numbers"))
    g_whole = S.fcc64_graphs(3, S.FCC_SEED + 1)
    g_shard = S.fcc64_shard(1, 3, 1)
    assert len(g_shard) == 1 and all(torch.equal(g_whole[1][k], g_shard[0][k]) for k in ("pos", "edge_index", "atomic_numbers"))
    one = S.fcc64_shard(0, 1, 2)
    ref = S.fcc64_graphs(2)
    assert all(torch.equal(a["pos"], b["pos"]) for a, b in zip(one, ref))
    src = inspect.getsource(bench.run_rank)
    assert "fcc64_shard(rank, world, B)" in src and "FCC_SEED + rank" not in src


def test_bench_rehearsal_flags(monkeypatch):
    """--backend gloo --share-gpu: the N > 1 branch of bench.py rehearsed on one device (RCCL refuses two ranks per GPU);
    the flags reach the ranks through the child launch, and --share-gpu without gloo is refused before any GPU call."""
    import importlib
    import subprocess
    import sys

    import bench

    importlib.reload(bench)
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.setdefault("cmd", cmd) and 0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MATTEN_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--backend", "gloo", "--share-gpu", "--no-extras"])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["cmd"][-6:] == ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--no-extras"]
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--share-gpu"])
    with pytest.raises(SystemExit, match="gloo"):
        bench.main()


def test_reference_training_script_call_sequence_under_the_matten_alias(golden_dir):
    """scripts/train_materials_tensor.py:11-14,34-52 of the reference, line by line, with `import matten` resolving to
    this package: data module -> get_to_model_info -> ScalarTensorModel(tasks=TensorRegressionTask(...), ...) ->
    configure_optimizers (what Trainer.fit calls first).  The forward itself needs the MI355X (tests/test_gpu_training)."""
    import matten
    import matten_amd
    from matten.dataset.structure_scalar_tensor import TensorDataModule
    from matten.log import set_logger
    from matten.model_factory.task import TensorRegressionTask
    from matten.model_factory.tfn_scalar_tensor import ScalarTensorModel
    from common import LMAX2

    assert ScalarTensorModel is matten_amd.model_factory.tfn_scalar_tensor.ScalarTensorModel
    assert matten.predict.predict is matten_amd.predict.predict
    import matten.nn.conv as c1
    import matten_amd.nn.conv as c2
    assert c1 is c2
    set_logger("ERROR")
    config = {
        "data": {"root": golden_dir, "tensor_target_name": "elastic_tensor_full",
                 "trainset_filename": "example_crystal_elasticity_tensor_n100.json",
                 "valset_filename": "example_crystal_elasticity_tensor_n100.json",
                 "testset_filename": "example_crystal_elasticity_tensor_n100.json",
                 "r_cut": 5.0, "reuse": False, "loader_kwargs": {"batch_size": 32, "shuffle": True}},
        "model": dict(LMAX2),
        "optimizer": {"class_path": "torch.optim.Adam", "init_args": {"lr": 0.01, "weight_decay": 0.00001}},
        "lr_scheduler": {"class_path": "torch.optim.lr_scheduler.ReduceLROnPlateau",
                         "init_args": {"mode": "min", "factor": 0.5, "patience": 50, "verbose": True}},
    }
    dm = TensorDataModule(**config["data"])
    dm.prepare_data()
    dm.setup()
    info = dm.get_to_model_info()
    assert len(info["allowed_species"]) == 73 and abs(info["average_num_neighbors"] - 30.4017) < 1e-3
    model = ScalarTensorModel(
        tasks=TensorRegressionTask(name=config["data"]["tensor_target_name"]),
        backbone_hparams=config["model"],
        dataset_hparams=info,
        optimizer_hparams=config["optimizer"],
        lr_scheduler_hparams=config["lr_scheduler"],
    )
    assert model.hparams["dataset_hparams"]["allowed_species"] == info["allowed_species"]
    cfg = model.configure_optimizers()
    assert isinstance(cfg["optimizer"], torch.optim.Adam) and cfg["monitor"] == "val/score"
    assert isinstance(cfg["lr_scheduler"], torch.optim.lr_scheduler.ReduceLROnPlateau)
    assert cfg["optimizer"].defaults["lr"] == 0.01 and cfg["optimizer"].defaults["weight_decay"] == 0.00001
    batch = next(iter(dm.train_dataloader()))
    assert batch["elastic_tensor_full"].shape == (32, 21) and batch["ptr"].shape == (33,)
    # the loss / metric half of shared_step on stand-in predictions (the decode half runs on the GPU)
    labels = {"elastic_tensor_full": batch["elastic_tensor_full"]}
    preds = {"elastic_tensor_full": batch["elastic_tensor_full"] + 0.5}
    individual, total = model.compute_loss(preds, labels)
    assert abs(float(total) - 0.25) < 1e-6 and set(individual) == {"elastic_tensor_full"}
    model.update_metrics(preds, labels, "val")
    model.on_validation_epoch_end()
    assert abs(float(model.logged["val/score"]) - 0.5) < 1e-6   # mean absolute error = the monitored score


def test_agg_linear_plan_reproduces_lin2_on_the_host():
    """plan.plan_agg_linear is a contract between two kernels: matten_tp_fused writes the neighbour sums component-major
    through the group entries' (first float, component stride) pairs, matten_agg_linear walks the row by the block list
    and multiplies with A fragments gathered from the flat lin2 weight.  Emulated here in numpy, table by table, against
    the oracle's FullyConnectedTensorProduct(agg_mul_ir, one_hot(species)): every channel of every path lands in exactly
    one slot, pad slots are never read as data, wide irreps split into table rows add up, the fragment layout matches the
    kernel's operand indexing."""
    from common import PAPER
    from matten_amd import plan as mplan
    from matten_amd.model_factory.tfn_scalar_tensor import create_model
    from oracle.e3nn_lite import o3 as ro3

    S = 3
    model = create_model(dict(PAPER), {"allowed_species": [13, 29, 79], "average_num_neighbors": 18.0})
    convs = [m for m in model.modules() if type(m).__name__ == "PointConv"]
    rng = np.random.default_rng(0)
    for conv in (convs[1], convs[3]):
        uvu, lp = conv.tp.plan, conv.lin2.plan
        ap = mplan.plan_agg_linear(uvu, S, lp.irreps_out)
        assert ap is not None and ap.ld % 32 == 0 and ap.n_chunks * 16 <= ap.ld
        N = 5
        agg = rng.standard_normal((N, uvu.d_mid)).astype(np.float64)          # reference layout: per path [u][k]
        species = rng.integers(0, S, N)
        w = rng.standard_normal(lp.weight_numel)
        # --- what tp_fused's epilogue does with the new entries: out_off[c] + u + k * t_off[c] ---
        row = np.full((N, ap.ld), np.nan)                                      # NaN: a slot that is read must have been written
        ent = ap.entries
        written = np.zeros(ap.ld, dtype=int)
        for e in range(len(ent)):
            mul = int(uvu.group_entry_mul[e])    # channels of the record OWN block (merged entries: 2 + 2 over two records)
            for c, pi in uvu.group_entry_paths[e].items():
                pth = uvu.paths[pi]
                d3 = 2 * pth.l3 + 1
                o, ks = int(ent[e][20 + c]), int(ent[e][8 + c])
                for u in range(mul):
                    for k in range(d3):
                        row[:, o + u + k * ks] = agg[:, pth.out_off + (uvu.group_entry_u0[e] + u) * d3 + k]
                        written[o + u + k * ks] += 1
        assert written.max() == 1 and written.sum() == uvu.d_mid               # a bijection onto the used slots
        # --- what agg_linear does: blocks -> chunks -> MFMA steps with A[t][mt][g][c][s] ---
        wtab = np.where(ap.gather >= 0, w[np.clip(ap.gather, 0, None)] * ap.scale[None, :], 0.0)
        out = np.zeros((N, ap.d_out))
        seen_chunks = 0
        for (chunk, info, t0, _) in ap.blocks.tolist():
            n, k, ii = info & 255, (info >> 12) & 255, (info >> 20) & 4095
            c0, T, K, packed, a_off, out_off, mo, k0 = ap.io_table[ii].tolist()
            d3, n_mt, cw, kk = packed & 255, (packed >> 8) & 255, (packed >> 16) & 255, (packed >> 24) & 255
            assert mo * kk <= mplan.AGG_STAGE_W and n_mt <= mplan.AGG_MAX_MT and 1 <= n <= mplan.AGG_BLOCK
            assert k0 <= k < k0 + kk <= d3   # the row's component range (a wide irrep is cut by component: no chunk read twice)
            assert chunk == c0 + k * T + t0
            for i in range(n):
                t = t0 + i
                seen_chunks += 1
                for g in range(4):
                    for s_ in range(4):
                        slot = 16 * t + 4 * g + s_
                        if slot >= K:
                            continue                                           # masked by the kernel (select, not multiply)
                        b = row[:, 16 * (chunk + i) + 4 * g + s_]
                        assert not np.isnan(b).any()
                        for mt in range(n_mt):
                            for c in range(min(cw, 16)):
                                v = 16 * mt + c
                                if v >= mo:
                                    continue
                                a_idx = a_off + ((((t * n_mt + mt) * 4 + g) * cw + c) * 4 + s_)
                                out[:, out_off + v * d3 + k] += wtab[species, a_idx] * b
        ref = ro3.FullyConnectedTensorProduct(str(uvu.irreps_out), f"{S}x0e", str(lp.irreps_out)).double()
        with torch.no_grad():
            ref.weight.copy_(torch.as_tensor(w))
            want = ref(torch.as_tensor(agg), torch.nn.functional.one_hot(torch.as_tensor(species), S).double()).numpy()
        assert np.allclose(out, want, rtol=0, atol=1e-6 * np.abs(want).max()), np.abs(out - want).max()   # (scale is fp32)


def test_alternative_coupling_groups_are_taken_only_by_blocks_they_cover():
    """plan.groups_for_block: an input block runs on ONE alternative group when that group holds all of its couplings (the
    even-l3 couplings of an odd- / even-parity block: what a layer cut down to 0e + 2e + 4e outputs keeps), on the regular
    groups otherwise; regular groups partition every coupling of their degree, alternatives come after them in the kind
    numbering, and the generated header instantiates exactly these kinds."""
    import re
    from matten_amd import plan as mplan

    for l1, groups in mplan.TP_GROUPS.items():
        n_reg = mplan.TP_GROUPS_REGULAR[l1]
        regular = [c for g in groups[:n_reg] for c in g]
        every = [(l2, l3) for l2 in range(5) for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1)]
        assert sorted(regular) == sorted(every)                       # a partition of the degree's couplings
        assert len(groups) <= mplan.TP_KIND_STRIDE
        assert mplan.groups_for_block(l1, every) == list(enumerate(groups[:n_reg]))
        for gi in range(n_reg, len(groups)):
            alt = groups[gi]
            small = all(max(c) <= 2 for c in alt)                              # the lmax-2 lists: every coupling of degrees <= 2
            assert small or (all(l3 % 2 == 0 for _, l3 in alt) and len({l2 % 2 for l2, _ in alt}) == 1)   # even l3, one parity of l2
            assert mplan.groups_for_block(l1, alt) == [(gi, alt)]
            sub = mplan.groups_for_block(l1, alt[:1])                          # a subset is covered too: by the shortest list
            assert len(sub) == 1 and sub[0][0] >= n_reg and set(alt[:1]) <= set(sub[0][1]) and len(sub[0][1]) <= len(alt)
            other = next(c for c in every if c not in alt)
            assert mplan.groups_for_block(l1, alt + [other]) == list(enumerate(groups[:n_reg]))
    header = open(os.path.join(ROOT, "matten_amd", "csrc", "cg_gen.h")).read()
    listed = re.search(r"#define MATTEN_FOR_EACH_GROUP\(X\) (.*)", header).group(1)
    want = " ".join(f"X({l1}, {gi})" for l1, g in mplan.TP_GROUPS.items() for gi in range(len(g)))
    assert listed.strip() == want, "cg_gen.h is stale: python matten_amd/csrc/gen_cg.py > matten_amd/csrc/cg_gen.h"


def test_inline_asm_mfma_objects_keep_their_hazard_distance(tmp_path):
    """species_linear.hip / species_linear_rows.hip issue v_mfma through inline asm, which the compiler's hazard recogniser
    cannot see: only their own s_nop pads (and -mllvm -simplifycfg-sink-common=false, csrc/Makefile) keep a store from reading an
    accumulator too early.  tools/isa_mfma_hazard.py walks the CFG of the gfx950 ISA of the BUILT objects and demands LLVM's own
    distance (passes + 2 = 10 wait states for the 8-pass fp32 instruction) from every v_mfma to every memory instruction that
    takes one of its destination registers -- so a ROCm upgrade that reschedules these kernels fails here, at build time,
    instead of in selfcheck.py on a GPU.  Calibration: the objects whose matrix instructions are builtins (scheduled BY that
    recogniser) pass at 10 and fail at 11; negative controls: a kernel without the drain, and hand-made instruction lists."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mfma_hazard as isa

    csrc = os.path.join(ROOT, "matten_amd", "csrc")
    objs = ["species_linear", "species_linear_rows", "agg_linear"]
    subprocess.run(["make", "-C", csrc] + [f"build/{o}.o" for o in objs], check=True, capture_output=True)
    for o, min_mfma in (("species_linear", 1000), ("species_linear_rows", 50)):
        findings, n_mfma, _ = isa.check_object(os.path.join(csrc, "build", f"{o}.o"))
        assert n_mfma >= min_mfma, (o, n_mfma)          # the disassembly was really found and parsed
        assert not findings, "\n".join(findings[:10])
    # calibration on a compiler-scheduled object: exactly LLVM's distance, no more
    agg = os.path.join(csrc, "build", "agg_linear.o")
    assert not isa.check_object(agg, required=10)[0]
    assert isa.check_object(agg, required=11)[0]
    # negative control 1: the same inline-asm form without the drain in front of the store
    src = tmp_path / "bad.hip"
    src.write_text('''#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void bad(const float* a, const float* b, f32x4* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float x = a[threadIdx.x], y = b[threadIdx.x];
    asm volatile("s_nop 1\\n\\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#ifdef DRAIN
    asm volatile("s_nop 15" ::: "memory");
#endif
    out[threadIdx.x] = acc;
}
''')
    for flags, want_bad in (([], True), (["-DDRAIN"], False)):
        obj = str(tmp_path / f"bad{len(flags)}.o")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", obj] + flags, check=True, capture_output=True)
        findings, n_mfma, _ = isa.check_object(obj)
        assert n_mfma == 1 and bool(findings) == want_bad, (flags, findings)
    # negative control 2: hand-made lists -- the short path is found through a loop's back edge, and an s_nop counts N + 1
    loop = [(0x00, "v_mov_b32_e32", "v9, 0"),
            (0x04, "global_store_dwordx4", "v[4:5], v[0:3], off"),           # loop head: reached 2 wait states behind the MFMA
            (0x0c, "s_nop", "15"),
            (0x10, "v_mfma_f32_16x16x4_f32", "v[0:3], v8, v9, v[0:3]"),
            (0x18, "s_cmp_lg_u32", "s0, 0"),
            (0x1c, "s_cbranch_scc1", str(65536 - 7)),                        # back to 0x04
            (0x20, "s_nop", "8"),
            (0x24, "s_endpgm", "")]
    f = isa.check_function("loop", loop)
    assert len(f) == 1 and "0x4" in f[0] and "v0 2 wait states" in f[0], f
    ok = [(a, m, ("15" if (m == "s_nop" and a == 0x20) else o)) for a, m, o in loop]
    ok[5] = (0x1c, "s_cbranch_scc1", "1")                                    # forward over the s_nop to s_endpgm: no store behind it
    assert not isa.check_function("ok", ok)
    spill = [(0x00, "v_mfma_f32_16x16x4_f32", "v[0:3], v8, v9, v[0:3]"), (0x08, "s_nop", "8"),
             (0x0c, "scratch_store_dword", "off, v2, off offset:8"), (0x14, "s_endpgm", "")]
    assert len(isa.check_function("spill", spill)) == 1                      # 9 < 10
    spill[1] = (0x08, "s_nop", "9")
    assert not isa.check_function("spill", spill)


class _TinyTaskModel(torch.nn.Module):
    """stand-in with the product models' call convention: model(batch, task_name=) -> ({task: [B, 21]}, labels)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.lin = torch.nn.Linear(7, 21)
        self.register_buffer("running", torch.zeros(3))

    def forward(self, batch, task_name="elastic_tensor_full"):
        return {task_name: self.lin(batch["x"])}, None


def _ddp_worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matten_amd.parallel import DataParallelStep, shard_bounds

    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(5, 7, generator=g), torch.randn(5, 21, generator=g)
    model = _TinyTaskModel()
    if rank == 1:   # a rank that starts elsewhere is pulled onto rank 0's parameters by the constructor's broadcast
        with torch.no_grad():
            model.lin.weight.add_(1.0)
    model.running.fill_(float(rank))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    dp = DataParallelStep(model, opt, lambda preds, t: torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t))
    lo, hi = shard_bounds(5, rank, world)          # 3 + 2: uneven shards
    losses = [float(dp.step({"x": x[lo:hi]}, y[lo:hi], 5)) for _ in range(3)]
    model.running.fill_(float(rank) + 1.0)
    dp.sync_buffers()
    torch.save({"w": model.lin.weight.detach().clone(), "b": model.lin.bias.detach().clone(), "losses": losses,
                "running": model.running.clone()}, os.path.join(tmp, f"ddp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_equals_the_full_batch_step(tmp_path):
    """parallel.DataParallelStep over two gloo ranks with uneven shards (3 + 2 samples): local mean loss weighted by
    n_local / n_global, ONE all-reduce of the gradients, optimiser step == the full-batch step of a single process, three steps
    in a row; both ranks end on the same parameters bit for bit; sync_buffers averages the floating-point buffers."""
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() + 77) % 2000
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(5, 7, generator=g), torch.randn(5, 21, generator=g)
    ref = _TinyTaskModel()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    want_losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(ref(dict(x=x))[0]["elastic_tensor_full"], y)
        loss.backward()
        opt.step()
        want_losses.append(float(loss))
    got = [torch.load(os.path.join(tmp_path, f"ddp{r}.pt")) for r in range(2)]
    assert torch.equal(got[0]["w"], got[1]["w"]) and torch.equal(got[0]["b"], got[1]["b"])
    assert torch.allclose(got[0]["w"], ref.lin.weight.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(got[0]["b"], ref.lin.bias.detach(), rtol=1e-5, atol=1e-6)
    assert np.allclose(got[0]["losses"], want_losses, rtol=1e-5) and got[0]["losses"] == got[1]["losses"]
    assert torch.equal(got[0]["running"], torch.full((3,), 1.5)) and torch.equal(got[1]["running"], got[0]["running"])
