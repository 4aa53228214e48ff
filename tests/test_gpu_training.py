"""
Training step on the GPU (BASELINE.json configs[3] / SURVEY.md 8d config 4: lmax=2, 3 gated blocks,
BatchNorm in training mode, MSE in irreps space, Adam) against the CPU oracle's autograd on identical
crystals and weights: forward values, every parameter gradient, BatchNorm running statistics and the
parameters after one Adam step.
"""
import copy
import os

import numpy as np
import pytest
import torch

from common import LMAX2, PAPER, build_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, rtol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1e-12, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _graphs(golden_dir, n=10):
    from matten_amd.data.graph import average_num_neighbors, crystal_graph
    from oracle.matten_ref.data import structures_from_json

    structs = structures_from_json(os.path.join(golden_dir, "example_crystal_elasticity_tensor_n100.json"))[:n]
    graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in structs]
    species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
    return graphs, {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}


@pytest.mark.parametrize("hp_name", ["lmax2", "lmax2_batch32", "paper"])
def test_training_step_matches_oracle_autograd(golden_dir, hp_name):
    """"lmax2_batch32" is BASELINE configs[3] at its stated batch size (32 crystals of the n100 sample)"""
    from matten_amd.data.graph import collate

    hp = {"lmax2": LMAX2, "lmax2_batch32": LMAX2, "paper": PAPER}[hp_name]
    graphs, ds = _graphs(golden_dir, {"lmax2": 10, "lmax2_batch32": 32, "paper": 6}[hp_name])
    ref, model = build_pair(hp, ds, randomize_bn=True)
    ref.train()
    model.train()
    B = len(graphs)
    target = torch.randn(B, 21, generator=torch.Generator().manual_seed(7))

    # ---- oracle: forward, MSE, backward, Adam ----
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-5)
    out_r = ref.decode(collate(graphs))
    loss_r = torch.nn.functional.mse_loss(out_r, target)
    opt_r.zero_grad()
    loss_r.backward()
    grads_r = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}

    # ---- HIP path ----
    opt_m = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-5)
    preds, _ = model(collate(graphs, device=DEV))
    out_m = preds["elastic_tensor_full"]
    loss_m = torch.nn.functional.mse_loss(out_m, target.to(DEV))
    opt_m.zero_grad()
    loss_m.backward()

    _close(out_m, out_r, 5e-4, "train-mode forward [B,21]")
    _close(loss_m, loss_r, 1e-4, "loss")
    if hp_name.startswith("lmax2"):  # config 4: the 4e block of the head is unreachable and stays exactly zero
        assert torch.all(out_m[:, 12:] == 0)
    named = dict(model.named_parameters())
    assert set(grads_r) <= set(named)
    for k, g in grads_r.items():
        assert named[k].grad is not None, k
        _close(named[k].grad, g, 3e-3, f"grad {k}")
    # BatchNorm running statistics after the forward
    bufs_m = dict(model.named_buffers())
    for k, b in ref.named_buffers():
        if k.endswith(("running_mean", "running_var")):
            _close(bufs_m[k], b, 1e-4, k)

    opt_r.step()
    opt_m.step()
    # Adam's first step moves every element by lr * g' / (|g'| + eps) = lr * sign(g'), g' = g + weight_decay * p: an
    # element whose g' lies inside the rounding noise between two fp32 evaluations of the same sum over ~10^4 edges (the
    # gradients above agree to 3e-3 of the tensor's largest; measured up to 1.8e-3 for the first radial MLP) may move by
    # 2 lr either way in ANY two correct implementations -- which ones do depends on the summation order, e.g. on the
    # length of the CSR pieces the forward walks.  Compare where the reference gradient is well above that noise;
    # elsewhere only require that the step stayed within +- lr.
    for k, p in ref.named_parameters():
        if p.grad is not None:
            g = grads_r[k] + 1e-5 * (p.detach() + 0.0)      # (p is already stepped; the shift of 1e-2 * 1e-5 is far below the mask)
            solid = (g.abs() >= 1e-2 * grads_r[k].abs().max()).to(named[k].device)
            got, want = named[k].detach(), p.detach().to(named[k].device)
            _close(torch.where(solid, got, want), want, 2e-3, f"param after Adam {k}")
            assert ((got - want).abs() <= 2.0 * 1e-2 + 1e-6).all(), k


def test_eval_after_training_uses_running_stats(golden_dir):
    """model.eval() after a training forward: the fused inference path agrees with the oracle again."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 6)
    ref, model = build_pair(LMAX2, ds, randomize_bn=True)
    ref.train()
    model.train()
    with torch.no_grad():
        ref.decode(collate(graphs))
        model(collate(graphs, device=DEV))
    ref.eval()
    model.eval()
    with torch.no_grad():
        want = ref.decode(collate(graphs))
        got = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
    _close(got, want, 5e-4, "eval after train")


def test_hipgraph_training_step_follows_the_eager_trajectory(golden_dir):
    """matten_amd.graphs.GraphedTrainStep: forward + MSE + backward + Adam of config 4 captured once in a hipGraph and
    replayed; after several steps on alternating batches of equal shape the parameters follow the eager run (the
    backward's fp32 atomics make both runs reproducible only to rounding), a batch of another shape is refused."""
    from matten_amd.data.graph import collate
    from matten_amd.graphs import GraphedTrainStep

    graphs, ds = _graphs(golden_dir, 8)
    batches = [collate(graphs, device=DEV), collate(graphs[::-1], device=DEV)]   # same shapes, different order
    targets = [torch.randn(len(graphs), 21, device=DEV, generator=torch.Generator(device=DEV).manual_seed(s)) for s in (1, 2)]

    def make():
        _, m = build_pair(LMAX2, ds, randomize_bn=True)
        m.train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5, fused=True, capturable=True)

    def loss_fn(preds, t):
        return torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)

    warm = 2
    eager, opt_e = make()
    graphed, opt_g = make()
    before = {k: v.detach().clone() for k, v in graphed.state_dict().items()}
    step = GraphedTrainStep(graphed, opt_g, loss_fn, batches[0], targets[0], warmup=warm)
    # the constructor's warm-up steps are undone (parameters, BatchNorm running statistics, Adam state restored in place)
    for k, v in graphed.state_dict().items():
        assert torch.equal(v, before[k]), f"{k} changed by the warm-up"
    assert all(float(st["step"]) == 0.0 for st in opt_g.state.values())
    for i in range(6):
        b, t = batches[i % 2], targets[i % 2]
        le = loss_fn(eager(dict(b))[0], t)
        opt_e.zero_grad(); le.backward(); opt_e.step()
        lg = step.step(b, t)
        _close(lg, le, 2e-3, f"loss at step {i}")
    for (k, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        _close(pg, pe, 5e-3, f"param {k} after 6 replayed steps")
    small = collate(graphs[:3], device=DEV)
    with pytest.raises(ValueError, match="captured step takes"):
        step.step(small, targets[0][:3])
    # a replayed batch is trusted unless validate=True: then a malformed one raises like the eager path
    bad = dict(batches[0])
    bad["edge_index"] = bad["edge_index"].clone()
    bad["edge_index"][0, 0] = bad["pos"].shape[0] + 5
    step.step(bad, targets[0])
    with pytest.raises(IndexError, match="edge_index holds node ids outside"):
        step.step(bad, targets[0], validate=True)
    step.step(batches[0], targets[0], validate=True)


def test_hipgraph_forward_is_bit_identical_to_eager():
    from matten_amd.data import synthetic
    from matten_amd.data.graph import collate
    from matten_amd.graphs import GraphedForward

    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    _, model = build_pair(PAPER, ds, randomize_bn=True)
    a, b = collate(synthetic.fcc64_graphs(4), device=DEV), collate(synthetic.fcc64_graphs(4, seed=99), device=DEV)
    g = GraphedForward(model, a)
    with torch.no_grad():
        for batch in (a, b, a):
            want = model(dict(batch))[0]["elastic_tensor_full"]
            assert torch.equal(g(batch), want)
    # a replayed batch must hold exactly the tensors of the construction batch: one that carries its own CSR (device graph
    # builder) replayed without it -- or the other way round -- would run on a stale CSR (round-5 advisor finding)
    with_csr = dict(a, _amd_rowptr=torch.zeros(a["pos"].shape[0] + 1, dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError, match="adds .*_amd_rowptr"):
        g(with_csr)
    with pytest.raises(ValueError, match="lacks .*num_neigh"):
        g({k: v for k, v in a.items() if k != "num_neigh"})


def test_hipgraph_forward_above_a_million_edges():
    """1000 fcc-64 crystals = 1.15 M edges: more keys than rocPRIM's radix sort survives in a capture on this ROCm
    (docs/LAB_NOTES.md round 2).  The CSR of a sparse graph is built by counting, without that sort, so the capture works;
    a DENSE graph of that size is refused up front."""
    from matten_amd.data import synthetic
    from matten_amd.data.graph import collate
    from matten_amd.graphs import GraphedForward, _check_capturable

    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    _, model = build_pair(dict(LMAX2, num_layers=1), ds, randomize_bn=True)
    graphs = synthetic.fcc64_graphs(64)
    a = collate([graphs[i % 64] for i in range(1000)], device=DEV)
    b = collate([graphs[(i * 7 + 3) % 64] for i in range(1000)], device=DEV)
    assert a["edge_index"].shape[1] > 2**20
    g = GraphedForward(model, a)
    with torch.no_grad():
        for batch in (a, b, a):
            assert torch.equal(g(batch), model(dict(batch))[0]["elastic_tensor_full"])
    dense = {"edge_index": torch.zeros(2, 2_000_000, dtype=torch.int64), "pos": torch.zeros(1000, 3)}
    with pytest.raises(ValueError, match="radix sort"):
        _check_capturable(dense)


def test_training_script_loop_through_the_matten_alias(golden_dir):
    """scripts/train_materials_tensor.py:34-66 of the reference on the MI355X: data module, model with the training
    shell (shared_step / compute_loss / metrics / configure_optimizers), `Trainer.fit` + `Trainer.test` -- with this
    package's minimal Trainer standing in for Lightning's (not installed here).  The loss falls over a few steps."""
    from matten.dataset.structure_scalar_tensor import TensorDataModule
    from matten.model.trainer import Trainer
    from matten.model_factory.task import TensorRegressionTask
    from matten.model_factory.tfn_scalar_tensor import ScalarTensorModel

    dm = TensorDataModule(trainset_filename="example_crystal_elasticity_tensor_n100.json",
                          valset_filename="example_crystal_elasticity_tensor_n100.json",
                          testset_filename="example_crystal_elasticity_tensor_n100.json", root=golden_dir, r_cut=5.0,
                          tensor_target_name="elastic_tensor_full", tensor_target_scale=1e-2,
                          loader_kwargs={"batch_size": 32, "shuffle": True}, device=DEV)
    dm.prepare_data()
    dm.setup()
    torch.manual_seed(35)
    model = ScalarTensorModel(
        tasks=TensorRegressionTask(name="elastic_tensor_full"), backbone_hparams=dict(LMAX2),
        dataset_hparams=dm.get_to_model_info(),
        optimizer_hparams={"class_path": "torch.optim.Adam", "init_args": {"lr": 0.01, "weight_decay": 0.00001}},
        lr_scheduler_hparams={"class_path": "torch.optim.lr_scheduler.ReduceLROnPlateau",
                              "init_args": {"mode": "min", "factor": 0.5, "patience": 50}},
    ).to(DEV)
    trainer = Trainer(max_epochs=3)
    trainer.fit(model, datamodule=dm)
    h = trainer.history
    assert len(h) == 3 and all(np.isfinite(e["val/score"]) for e in h)
    assert h[-1]["train/total_loss"] < h[0]["train/total_loss"]
    assert h[-1]["val/score"] < h[0]["val/score"]                 # mean absolute error on the (same) validation file
    out = trainer.test(model, datamodule=dm)
    assert "metric_test/MeanAbsoluteError/elastic_tensor_full" in out[0]


# (48 and 842 columns: two kernels; 128-512: the one-kernel adjoint at 4 / 6 / 8 chunks per wave; 70 001 edges: two rounds
# per workgroup with a ragged last one)
@pytest.mark.parametrize("n_edges,W", [(1, 48), (37, 216), (200, 128), (5000, 432), (70001, 330), (4096 + 17, 842)])
def test_radial_mlp_kernels_forward_and_adjoint(n_edges, W):
    """matten_radial_mlp / matten_radial_mlp_bwd (fp32 MFMA, partial sums in a fixed order) against the plain formula of
    e3nn FullyConnectedNet([8, 32, 32, W], silu) differentiated by torch autograd in fp64: w, dW0, dW1, dW2; edges at and
    beyond the cutoff included, edge counts that leave ragged tiles / several partial-sum ranges; bitwise reproducible."""
    from matten_amd.nn.utils import RadialMLP
    from oracle.e3nn_lite.math import soft_one_hot_linspace

    g = torch.Generator().manual_seed(n_edges + W)
    mlp = RadialMLP([8, 32, 32, W]).to(DEV)
    lens = torch.rand(n_edges, generator=g, dtype=torch.float64) * 5.4 + 0.05
    if n_edges > 3:
        lens[1], lens[2] = 5.0, 6.5      # at / beyond the cutoff: zero embedding
    geom = torch.zeros(n_edges, 4, device=DEV)
    geom[:, 3] = lens.float().to(DEV)
    gout = torch.randn(n_edges, (W + 15) // 16 * 16, generator=g).to(DEV)

    w = mlp.forward_train(geom, 8, 0.0, 5.0)
    assert w.shape == gout.shape and torch.all(w[:, W:] == 0)
    (w * gout).sum().backward()
    got = [mlp.layer0.weight.grad.clone(), mlp.layer1.weight.grad.clone(), mlp.layer2.weight.grad.clone()]
    for p in mlp.parameters():
        p.grad = None
    w_again = mlp.forward_train(geom, 8, 0.0, 5.0)
    (w_again * gout).sum().backward()
    assert torch.equal(w, w_again)
    assert all(torch.equal(a, p.grad) for a, p in zip(got, (mlp.layer0.weight, mlp.layer1.weight, mlp.layer2.weight)))

    ws = [p.detach().cpu().double().requires_grad_(True) for p in (mlp.layer0.weight, mlp.layer1.weight, mlp.layer2.weight)]
    x = soft_one_hot_linspace(lens.float().double(), 0.0, 5.0, 8, "bessel", True) * 8**0.5
    c = mlp.act_cst
    h = torch.nn.functional.silu(x @ (ws[0] / 8**0.5)) * c
    h = torch.nn.functional.silu(h @ (ws[1] / 32**0.5)) * c
    want = h @ (ws[2] / 32**0.5)
    (want * gout[:, :W].cpu().double()).sum().backward()
    _close(w[:, :W], want, 2e-6, "radial weights")
    for name, a, b in zip(("dW0", "dW1", "dW2"), got, ws):
        _close(a, b.grad, 2e-5, name)


def test_training_step_with_bf16_edge_storage(golden_dir):
    """BASELINE configs[3] names bf16 storage.  Opt-in here for the two per-edge tensors that dominate a large batch's
    traffic (radial weights w[E, W] and dL/dw): stored bf16 (round to nearest even), computed and accumulated in fp32;
    everything per node and every parameter stays fp32.  Against the fp32 oracle: forward within 1e-2, gradients within
    1e-1 of the gradient's largest magnitude (measured worst 4.6e-2; bf16 carries 8 significant bits: 4e-3 per element over ~30
    edges per node), and the loss still falls."""
    from matten_amd import autograd as ag
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 32)
    ref, model = build_pair(LMAX2, ds, randomize_bn=True)
    ref.train()
    model.train()
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(7))
    out_r = ref.decode(collate(graphs))
    torch.nn.functional.mse_loss(out_r, target).backward()
    ag.set_edge_storage_dtype(torch.bfloat16)
    try:
        batch = collate(graphs, device=DEV)
        out_m = model(dict(batch))[0]["elastic_tensor_full"]
        loss = torch.nn.functional.mse_loss(out_m, target.to(DEV))
        loss.backward()
        _close(out_m, out_r, 1e-2, "bf16 edge storage: forward [B,21]")
        named = dict(model.named_parameters())
        worst = 0.0
        for k, p in ref.named_parameters():
            if p.grad is not None:
                g, w = named[k].grad.detach().cpu().double(), p.grad.double()
                worst = max(worst, (g - w).abs().max().item() / max(1e-12, w.abs().max().item()))
                _close(named[k].grad, p.grad, 1e-1, f"bf16 edge storage: grad {k}")
        print(f"bf16 edge storage: worst relative gradient error {worst:.2e}")
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        first = None
        for _ in range(8):
            l = torch.nn.functional.mse_loss(model(dict(batch))[0]["elastic_tensor_full"], target.to(DEV))
            first = first if first is not None else float(l)
            opt.zero_grad(); l.backward(); opt.step()
        assert float(l) < first
    finally:
        ag.set_edge_storage_dtype(torch.float32)


def test_flat_adam_follows_torch_adam(golden_dir):
    """matten_amd.optim.FlatAdam (one launch over one flat buffer) against torch.optim.Adam in fp64 on
    the same gradients: parameters after 5 steps, with weight decay; then a real training step of the model and a
    hipGraph capture of it (the step counter lives on the device)."""
    from matten_amd.data.graph import collate
    from matten_amd.graphs import GraphedTrainStep
    from matten_amd.optim import FlatAdam

    g = torch.Generator().manual_seed(3)
    shapes = [(7,), (3, 5), (1,), (33, 2), (64,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().double().clone()) for p in ps]
    opt = FlatAdam(ps, lr=1e-2, weight_decay=1e-5)
    opt_r = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-5)
    for it in range(5):
        grads = [torch.randn(s, generator=g) for s in shapes]
        opt.zero_grad()
        for p, r, gr in zip(ps, ref, grads):
            (p * gr.to(DEV)).sum().backward()
            r.grad = gr.double()
        opt.step()
        opt_r.step()
    for p, r in zip(ps, ref):
        _close(p, r, 1e-6, "FlatAdam parameter")
    assert all(p.data_ptr() == opt.flat_params.data_ptr() + 4 * o for p, o in zip(ps, opt._offs))

    graphs, ds = _graphs(golden_dir, 8)
    batch, target = collate(graphs, device=DEV), torch.randn(8, 21, device=DEV)

    def make():
        _, m = build_pair(LMAX2, ds, randomize_bn=True)
        return m.train()

    def loss_fn(preds, t):
        return torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)

    m1, m2 = make(), make()
    o1 = FlatAdam(m1.parameters(), lr=1e-2, weight_decay=1e-5)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-2, weight_decay=1e-5)
    for _ in range(3):
        for m, o in ((m1, o1), (m2, o2)):
            loss = loss_fn(m(dict(batch))[0], target)
            o.zero_grad()
            loss.backward()
            o.step()
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        _close(a, b, 5e-3, f"param {k} after 3 steps (FlatAdam vs torch Adam)")
    m3 = make()
    o3 = FlatAdam(m3.parameters(), lr=1e-2, weight_decay=1e-5)
    step = GraphedTrainStep(m3, o3, loss_fn, batch, target, warmup=2)
    l0 = float(step.step(batch, target))
    for _ in range(5):
        l = float(step.step(batch, target))
    assert l < l0 and float(o3.step_count) == 6.0


@pytest.mark.parametrize("graphed", [False, True])
def test_eval_after_flat_adam_steps_uses_the_updated_weights(golden_dir, graphed):
    """The Trainer.fit loop: eval (packs the weights: species-linear tables, radial MLP fragments, folded BatchNorm) ->
    optimiser steps through raw pointers (FlatAdam, eager or replayed from a hipGraph; BatchNorm running statistics
    updated in place by the training kernels) -> eval.  No parameter's ``_version`` moves in between, so the caches hang
    on the process-wide weights epoch (nn/_tables.py): the second eval must match the oracle evaluated with the model's
    CURRENT state, and differ from the first."""
    from matten_amd.data.graph import collate
    from matten_amd.graphs import GraphedTrainStep
    from matten_amd.optim import FlatAdam

    graphs, ds = _graphs(golden_dir, 8)
    ref, model = build_pair(LMAX2, ds, randomize_bn=True)
    batch, target = collate(graphs, device=DEV), torch.randn(8, 21, generator=torch.Generator().manual_seed(9)).to(DEV)

    def loss_fn(preds, t):
        return torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)

    with torch.no_grad():
        before = model(dict(batch))[0]["elastic_tensor_full"].clone()
    _close(before, ref.decode(collate(graphs)), 2e-5, "eval before training")
    model.train()
    opt = FlatAdam(model.parameters(), lr=1e-2, weight_decay=1e-5)
    if graphed:
        step = GraphedTrainStep(model, opt, loss_fn, batch, target, warmup=1)
        for _ in range(3):
            step.step(batch, target)
    else:
        for _ in range(3):
            loss = loss_fn(model(dict(batch))[0], target)
            opt.zero_grad()
            loss.backward()
            opt.step()
    model.eval()
    with torch.no_grad():
        after = model(dict(batch))[0]["elastic_tensor_full"].clone()
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref.load_state_dict(state, strict=False)
    with torch.no_grad():
        want = ref.eval().decode(collate(graphs))
    assert (after - before).abs().max().item() > 1e-3 * before.abs().max().item(), "the training steps changed nothing"
    _close(after, want, 2e-5, "eval after FlatAdam steps (stale weight packs?)")


def test_flat_adam_state_dict_round_trip():
    """load_state_dict restores the LIVE flat moments and step count (torch's loader would park copies that step()
    never reads): resume from this class's own state_dict and from torch.optim.Adam's, then follow torch.optim.Adam."""
    from matten_amd import _lib
    from matten_amd.optim import FlatAdam

    g = torch.Generator().manual_seed(4)
    shapes = [(5,), (4, 3), (1,), (17, 2)]

    def fresh(vals):
        return [torch.nn.Parameter(v.clone().to(DEV)) for v in vals]

    init = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(6)]

    def run(opt, ps, its):
        for it in its:
            opt.zero_grad()
            for p, gr in zip(ps, grads[it]):
                (p * gr.to(DEV)).sum().backward()
            opt.step()

    ps_t = fresh(init)
    opt_t = torch.optim.Adam(ps_t, lr=1e-2, weight_decay=1e-5)
    run(opt_t, ps_t, range(3))
    mid_params = [p.detach().clone() for p in ps_t]
    sd_torch = copy.deepcopy(opt_t.state_dict())
    run(opt_t, ps_t, range(3, 6))

    ps_a = fresh(init)
    opt_a = FlatAdam(ps_a, lr=1e-2, weight_decay=1e-5)
    run(opt_a, ps_a, range(3))
    sd_flat = copy.deepcopy(opt_a.state_dict())
    for sd, what in ((sd_flat, "own state_dict"), (sd_torch, "torch.optim.Adam state_dict")):
        ps = fresh(mid_params)
        opt = FlatAdam(ps, lr=1e-3)                       # hyper-parameters come back from the saved group too
        opt.load_state_dict(sd)
        assert float(opt.step_count) == 3.0 and opt.param_groups[0]["lr"] == 1e-2
        assert opt.state[ps[0]]["exp_avg"].data_ptr() == opt.exp_avg.data_ptr()
        run(opt, ps, range(3, 6))
        for p, r in zip(ps, ps_t):
            _close(p, r, 2e-6, f"parameter after resuming from {what}")
    ps = fresh(init)
    opt = FlatAdam(ps, lr=1e-2)
    ps[1].data = ps[1].data.clone()                        # what model.to() / .double() after construction does
    ps[1].grad = torch.zeros_like(ps[1])
    with pytest.raises(_lib.MattenHipError, match="flat buffer"):
        opt.step()


@pytest.mark.parametrize("irreps_in,irreps_out,S,N", [
    ("32x0e+16x1o+4x2e", "32x0e+16x1o+4x2e", 3, 7),          # a handful of rows: one slice, mostly idle row quads
    ("32x0e+16x1o+4x2e+2x3o+2x4e", "16x0e+8x1o+4x2e+2x3o+1x4e", 4, 301),
    ("70x0e+33x1o", "65x0e+17x1o", 2, 150),                   # more than 4 x 4 tiles per segment: two tile blocks
    ("32x0e+16x1e+16x1o+4x2e", "32x0e+16x1e+16x1o+4x2e", 2, 6000),   # large batch: row slices + ordered reduction
    ("8x0e+3x2e", "5x0e+2x2e", 1, 4000),                      # single species, sliced
    ("16x0e+8x1o", "8x0e+8x1o", 70, 3000),                    # more species than one wave scans at once, skewed, two empty
])
def test_species_linear_gradients_vs_oracle_autograd(irreps_in, irreps_out, S, N):
    """matten_species_linear_wgrad (fp32 MFMA over (row, component), fixed summation order, sliced rows for large
    batches) and the transposed-table dx against autograd through the oracle's FullyConnectedTensorProduct in fp64;
    two runs are bit-identical (no atomics)."""
    from matten_amd import ops
    from matten_amd.nn.utils import SpeciesLinear
    from oracle.e3nn_lite import o3 as ro3

    gen = torch.Generator(device=DEV).manual_seed(11)
    mod = SpeciesLinear(irreps_in, S, irreps_out).to(DEV)
    ref = ro3.FullyConnectedTensorProduct(irreps_in, f"{S}x0e", irreps_out).double()
    w = torch.randn(mod.weight.numel(), device=DEV, generator=gen)
    x = torch.randn(N, mod.plan.d_in, device=DEV, generator=gen, requires_grad=True)
    gy = torch.randn(N, mod.plan.d_out, device=DEV, generator=gen)
    species = torch.randint(0, S, (N,), device=DEV, generator=gen)
    if N > 100:
        species[: N // 3] = 0                                    # ragged species groups
    if S > 64:
        species[species == 5] = 6                                # species without rows (they still own an item: zeros)
        species[species == S - 1] = S - 2
        species[N // 3: N // 3 + 128] = 66                       # >= one full slice in the second scan chunk
    order, seg, _ = ops.group_by_key(species, S)
    with torch.no_grad():
        mod.weight.copy_(w)
        ref.weight.copy_(w.cpu().double())
    grads = []
    for _ in range(2):
        mod.weight.grad = None
        x.grad = None
        mod(x, (order, seg)).backward(gy)
        grads.append((mod.weight.grad.clone(), x.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    xr = x.detach().cpu().double().requires_grad_(True)
    ref(xr, torch.nn.functional.one_hot(species.cpu(), S).double()).backward(gy.cpu().double())
    _close(grads[0][0], ref.weight.grad, 3e-6, "dL/dweight")
    _close(grads[0][1], xr.grad, 3e-6, "dL/dx")


def test_training_gradients_are_bitwise_reproducible(golden_dir):
    """No atomics on the step: dL/dx of the tensor product is summed per source node in a fixed order (the CSR of the
    source column), weight gradients and the radial MLP's adjoint are ordered partial sums -- two evaluations of the same
    batch give bit-identical gradients (with MATTEN_TP_BWD_DX=atomic they differ in the last bits)."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 24)
    _, model = build_pair(LMAX2, ds, randomize_bn=True)
    model.train()
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(5)).to(DEV)
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        bn = copy.deepcopy({k: v.clone() for k, v in model.named_buffers()})
        preds, _ = model(collate(graphs, device=DEV))
        torch.nn.functional.mse_loss(preds["elastic_tensor_full"], target).backward()
        runs.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        with torch.no_grad():   # same BatchNorm running statistics for the second evaluation
            for k, v in model.named_buffers():
                v.copy_(bn[k])
    assert runs[0].keys() == runs[1].keys() and len(runs[0]) > 20
    diff = [k for k in runs[0] if not torch.equal(runs[0][k], runs[1][k])]
    assert not diff, diff


def test_instance_normalization_forward_and_gradients(golden_dir):
    """normalization='instance' = the reference's own graph-wise InstanceNorm (nn/utils.py:448-588): per-crystal
    statistics, every l = 0 channel centred and biased, no running averages -- evaluation and training forward against
    the oracle, then every parameter gradient (crystals of 1 to 10 atoms in one batch)."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 14)
    hp = dict(LMAX2, normalization="instance")
    ref, model = build_pair(hp, ds)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():   # non-trivial affine parameters on both sides
        for (k, p), (k2, q) in zip(ref.named_parameters(), model.named_parameters()):
            if ".norm.n." in k:
                assert k == k2
                v = (0.5 + torch.rand(p.shape, generator=gen)) if k.endswith("weight") else 0.1 * torch.randn(p.shape, generator=gen)
                p.copy_(v)
                q.copy_(v.to(DEV))
    assert any(".norm.n.weight" in k for k, _ in model.named_parameters())
    assert not any("running" in k for k, _ in model.named_buffers())
    with torch.no_grad():
        want = ref.decode(collate(graphs))
        got = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
    _close(got, want, 2e-4, "eval forward with instance normalisation")
    ref.train(), model.train()
    target = torch.randn(len(graphs), 21, generator=gen)
    out_r = ref.decode(collate(graphs))
    torch.nn.functional.mse_loss(out_r, target).backward()
    out_m = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
    torch.nn.functional.mse_loss(out_m, target.to(DEV)).backward()
    _close(out_m, out_r, 5e-4, "train-mode forward (same arithmetic as eval)")
    named = dict(model.named_parameters())
    n = 0
    for k, p in ref.named_parameters():
        if p.grad is not None:
            assert named[k].grad is not None, k
            _close(named[k].grad, p.grad, 3e-3, f"grad {k}")
            n += 1
    assert n > 25


@pytest.mark.parametrize("N", [2047, 2048, 5003])
def test_batchnorm_training_reductions_at_large_row_counts(N):
    """matten_bn_train_fwd / _bwd on both sides of the row count where the reductions switch from one workgroup per
    channel to the two-stage column form (16-row blocks, records merged per channel in a fixed order): statistics,
    output, running averages and all three gradients against the e3nn BatchNorm arithmetic written out in fp64; the
    0e columns sit on an offset 50x their spread (the pairwise mean / squared-deviation merge must not cancel)."""
    from matten_amd.nn.utils import _IrrepBatchNorm
    from matten_amd.o3 import Irreps

    irreps = Irreps("8x0e+4x0o+5x1o+3x2e")
    gen = torch.Generator(device=DEV).manual_seed(5)
    bn = _IrrepBatchNorm(irreps).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(bn.weight.shape, device=DEV, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(bn.bias.shape, device=DEV, generator=gen))
    x = torch.randn(N, irreps.dim, device=DEV, generator=gen)
    x[:, :8] += 50.0
    x.requires_grad_(True)
    gy = torch.randn(N, irreps.dim, device=DEV, generator=gen)
    outs = []
    for _ in range(2):
        bn.running_mean.zero_(), bn.running_var.fill_(1.0)
        x.grad = bn.weight.grad = bn.bias.grad = None
        y = bn.forward_train(x)
        y.backward(gy)
        outs.append([t.detach().clone() for t in (y, x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)   # fixed summation order
    # e3nn BatchNorm (training mode, affine, "component" normalisation, mean reduce), fp64
    xr = x.detach().cpu().double().requires_grad_(True)
    w = bn.weight.detach().cpu().double().requires_grad_(True)
    b = bn.bias.detach().cpu().double().requires_grad_(True)
    cols, means, nus = [], [], []
    off = ic = ib = 0
    for mul, ir in irreps:
        blk = xr[:, off:off + mul * ir.dim].reshape(N, mul, ir.dim)
        if ir.is_scalar():
            mu = blk.mean(dim=(0, 2))
            blk = blk - mu[None, :, None]
            means.append(mu)
        nu = blk.pow(2).mean(dim=(0, 2))
        nus.append(nu)
        blk = blk * (w[ic:ic + mul] / (nu + bn.eps).sqrt())[None, :, None]
        if ir.is_scalar():
            blk = blk + b[ib:ib + mul][None, :, None]
            ib += mul
        cols.append(blk.reshape(N, -1))
        off += mul * ir.dim
        ic += mul
    yr = torch.cat(cols, 1)
    yr.backward(gy.cpu().double())
    y, dx, dw, db, rm, rv = outs[0]
    _close(y, yr, 2e-5, "y")   # (x - mean) of the offset columns carries the fp32 rounding of x itself: 50 * 2^-24 / 1
    _close(dx, xr.grad, 2e-5, "dL/dx")
    _close(dw, w.grad, 2e-5, "dL/dweight")
    _close(db, b.grad, 2e-5, "dL/dbias")
    _close(rm, 0.1 * torch.cat(means), 1e-6, "running_mean")
    _close(rv, 0.9 + 0.1 * torch.cat(nus), 1e-5, "running_var")


@pytest.mark.parametrize("normalization", ["batch", "instance", None])
def test_norm_activation_forward_and_gradients(golden_dir, normalization):
    """nonlinearity_type='norm' = e3nn NormActivation as the reference configures it (nn/utils.py:142-150): every
    channel scaled by silu(|x|) / |x|.  Evaluation forward (with the eval-mode BatchNorm folded into the kernel),
    training forward and every parameter gradient against the oracle."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 9)
    hp = dict(LMAX2, nonlinearity_type="norm", normalization=normalization)
    ref, model = build_pair(hp, ds, randomize_bn=True)
    assert type(model.backbone.layer0_convnet.act.plan).__name__ == "NormActPlan"
    with torch.no_grad():
        want = ref.decode(collate(graphs))
        got = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
    _close(got, want, 2e-4, "eval forward with the norm activation")
    ref.train(), model.train()
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(4))
    out_r = ref.decode(collate(graphs))
    torch.nn.functional.mse_loss(out_r, target).backward()
    out_m = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
    torch.nn.functional.mse_loss(out_m, target.to(DEV)).backward()
    _close(out_m, out_r, 5e-4, "train-mode forward")
    named = dict(model.named_parameters())
    n = 0
    for k, p in ref.named_parameters():
        if p.grad is not None:
            assert named[k].grad is not None, k
            _close(named[k].grad, p.grad, 3e-3, f"grad {k}")
            n += 1
    assert n > 20


@pytest.mark.parametrize("reduce", ["min", "max"])
def test_min_max_pooling_forward_and_gradients(golden_dir, reduce):
    """reduce = min | max (nn/nodewise.py:131,142-148: torch_scatter.scatter with that reduction): forward and all
    gradients against the oracle (the gradient of a pooled value goes to the atom it came from)."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 8)
    ref, model = build_pair(dict(LMAX2, reduce=reduce), ds, randomize_bn=True)
    with torch.no_grad():
        _close(model(collate(graphs, device=DEV))[0]["elastic_tensor_full"], ref.decode(collate(graphs)), 2e-4, reduce)
    ref.train(), model.train()
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(4))
    torch.nn.functional.mse_loss(ref.decode(collate(graphs)), target).backward()
    torch.nn.functional.mse_loss(model(collate(graphs, device=DEV))[0]["elastic_tensor_full"], target.to(DEV)).backward()
    named = dict(model.named_parameters())
    for k, p in ref.named_parameters():
        if p.grad is not None:
            _close(named[k].grad, p.grad, 3e-3, f"grad {k}")


def test_fused_training_path_with_the_paper_model(golden_dir, monkeypatch):
    """The l = 3, 4 instantiations of the fused training path (forward on matten_tp_fused, w-free adjoint
    matten_tp_backward_lit_wfree<4>, blocks whose paths take the LDS tile in rounds) on the paper's lmax-4 model: every gradient
    against the oracle's autograd and against the path kernels' (materialised w), and equal to its own repeat bit for bit."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 8)
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(11))
    grads = {}
    for mode in ("fused", "paths", "fused"):
        monkeypatch.setenv("MATTEN_TRAIN_TP", mode)
        ref, model = build_pair(PAPER, ds, randomize_bn=True)
        ref.train(), model.train()
        out = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
        torch.nn.functional.mse_loss(out, target.to(DEV)).backward()
        g = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        if mode in grads:
            assert all(torch.equal(g[k], grads[mode][k]) for k in g), "fused-path gradients differ between two evaluations"
            continue
        torch.nn.functional.mse_loss(ref.decode(collate(graphs)), target).backward()
        for k, p in ref.named_parameters():
            if p.grad is not None:
                _close(g[k], p.grad, 3e-3, f"[{mode}] grad {k}")
        grads[mode] = g
    for k in grads["fused"]:
        _close(grads["fused"][k], grads["paths"][k], 2e-3, f"fused vs path-kernel grad {k}")


def test_training_on_the_production_tensor_product_kernel(golden_dir, monkeypatch):
    """MATTEN_TRAIN_TP=fused: the training forward runs matten_tp_fused (w[E, W] never materialised in the forward, nothing
    per-edge saved for the backward); the backward re-evaluates w per layer.  Forward, every gradient and the parameters
    against the oracle; same gradients as the default path (the forward differs by the fp16-split rounding)."""
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 12)
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(7))
    grads = {}
    for mode in ("fused", "paths"):
        monkeypatch.setenv("MATTEN_TRAIN_TP", mode)
        ref, model = build_pair(LMAX2, ds, randomize_bn=True)
        ref.train(), model.train()
        torch.nn.functional.mse_loss(ref.decode(collate(graphs)), target).backward()
        out = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
        torch.nn.functional.mse_loss(out, target.to(DEV)).backward()
        named = dict(model.named_parameters())
        for k, p in ref.named_parameters():
            if p.grad is not None:
                _close(named[k].grad, p.grad, 3e-3, f"[{mode}] grad {k}")
        grads[mode] = {k: p.grad.clone() for k, p in model.named_parameters()}
    for k in grads["fused"]:
        _close(grads["fused"][k], grads["paths"][k], 2e-3, f"fused vs default grad {k}")
    # the fused mode's adjoint re-evaluates w on the matrix cores inside its workgroups (matten_tp_backward_lit_wfree, split
    # fp16 like the forward); with autograd.W_FREE_ADJOINT off it reads a w[E, W] that matten_radial_mlp wrote on the fp32
    # matrix instruction: same gradients to the split's rounding, and the w-free ones equal their own repeat bit for bit
    from matten_amd import autograd as mag

    monkeypatch.setenv("MATTEN_TRAIN_TP", "fused")
    for flag in (False, True):
        monkeypatch.setattr(mag, "W_FREE_ADJOINT", flag)
        _, model = build_pair(LMAX2, ds, randomize_bn=True)
        model.train()
        out = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
        torch.nn.functional.mse_loss(out, target.to(DEV)).backward()
        for k, p in model.named_parameters():
            if flag:
                assert torch.equal(p.grad, grads["fused"][k]), k
            else:
                _close(p.grad, grads["fused"][k], 1e-3, f"w-free vs materialised-w adjoint grad {k}")
    # the default ("auto") takes the fused forward from nn.utils.TRAIN_FUSED_MIN_EDGES edges on, the path kernels below
    from matten_amd.nn import utils as nnu

    monkeypatch.delenv("MATTEN_TRAIN_TP")
    for min_edges, like in ((0, "fused"), (1 << 40, "paths")):
        monkeypatch.setattr(nnu, "TRAIN_FUSED_MIN_EDGES", min_edges)
        _, model = build_pair(LMAX2, ds, randomize_bn=True)
        model.train()
        out = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
        torch.nn.functional.mse_loss(out, target.to(DEV)).backward()
        assert all(torch.equal(p.grad, grads[like][k]) for k, p in model.named_parameters()), like
    # capturable: the kernel's operands are derived by kernels, nothing is read on the host
    from matten_amd.graphs import GraphedTrainStep

    monkeypatch.setenv("MATTEN_TRAIN_TP", "fused")
    _, model = build_pair(LMAX2, ds, randomize_bn=True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
    batch, tgt = collate(graphs, device=DEV), target.to(DEV)
    gs = GraphedTrainStep(model, opt, lambda preds, t: torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t), batch, tgt,
                          warmup=2)
    l0 = float(gs.step(batch, tgt))
    for _ in range(5):
        l1 = float(gs.step(batch, tgt))
    assert np.isfinite(l1) and l1 < l0


def test_fused_training_with_an_input_block_wider_than_256_channels(golden_dir, monkeypatch):
    """A workgroup of matten_tp_backward_lit_wfree holds 256 / (lanes per edge) edges: an input block of more than 256 channels
    has none (round-5 advisor finding: epw = 0, a division by zero, channels >= 256 never visited -- silently wrong gradients).
    Such a layer keeps the fused forward and takes the materialised-w adjoint; the library refuses the wide block outright."""
    from matten_amd import _lib, ops
    from matten_amd.data.graph import collate

    graphs, ds = _graphs(golden_dir, 4)
    hp = dict(LMAX2, conv_layer_irreps="288x0e+8x0o+16x1o+4x1e+4x2e", num_layers=2)
    target = torch.randn(len(graphs), 21, generator=torch.Generator().manual_seed(3))
    grads = {}
    for mode in ("fused", "paths"):
        monkeypatch.setenv("MATTEN_TRAIN_TP", mode)
        ref, model = build_pair(hp, ds, randomize_bn=True)
        ref.train(), model.train()
        convs = [m for m in model.modules() if hasattr(m, "plan") and hasattr(m.plan, "bw_max_mul")]
        assert max(m.plan.bw_max_mul for m in convs) == 288
        torch.nn.functional.mse_loss(ref.decode(collate(graphs)), target).backward()
        out = model(collate(graphs, device=DEV))[0]["elastic_tensor_full"]
        torch.nn.functional.mse_loss(out, target.to(DEV)).backward()
        named = dict(model.named_parameters())
        for k, p in ref.named_parameters():
            if p.grad is not None:
                _close(named[k].grad, p.grad, 3e-3, f"[{mode}] grad {k}")
        grads[mode] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    for k in grads["fused"]:
        _close(grads["fused"][k], grads["paths"][k], 2e-3, f"fused vs path-kernel grad {k}")
    # the entry point itself: max_mul > 256 and a weight tile below the narrowest block's 2048 floats are argument errors
    wide = next(m for m in convs if m.plan.bw_max_mul == 288)
    p = wide.plan
    E, N = 64, 8
    x = torch.randn(N, p.d_in, device=DEV)
    args = dict(sh_sorted=torch.randn(E, 32, device=DEV), src_sorted=torch.zeros(E, dtype=torch.int32, device=DEV),
                dst_sorted=torch.zeros(E, dtype=torch.int32, device=DEV))
    wfree = (torch.zeros(E, 2, 32, dtype=torch.float16, device=DEV), torch.zeros(p.bw_a_tiles * 64 * 16, dtype=torch.float16, device=DEV),
             torch.ones(len(p.bw_paths), device=DEV))
    call = lambda **kw: ops.tp_backward_lit(x, None, args["sh_sorted"], args["src_sorted"], args["dst_sorted"],
                                            wide._tables.get("bw_blocks", torch.device(DEV)), wide._tables.get("bw_paths", torch.device(DEV)),
                                            p.bw_sum_lanes, torch.randn(N, p.d_mid, device=DEV), 10.0, wfree=wfree,
                                            dw_shape=((E, 16 * ((p.weight_numel + 15) // 16)), torch.float32), max_l=p.bw_max_l, **kw)
    with pytest.raises(_lib.MattenHipError, match="EINVAL"):
        call(lds_floats=p.bw_wfree_lds_floats, max_mul=p.bw_max_mul)
    with pytest.raises(_lib.MattenHipError, match="EINVAL"):
        call(lds_floats=1024, max_mul=256)
