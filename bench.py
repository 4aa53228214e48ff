#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X message-passing hot path.

  python bench.py --gpus N --steps K --warmup W
  N > 1 (or --force-dist / MATTEN_FORCE_DIST=1 with N = 1): when not already running as a rank of torch.distributed.run
  (RANK / WORLD_SIZE in the environment: how the driver starts it), the process -- which has not touched the GPU --
  starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD and exits with
  its code; the ranks (one per GPU) use RCCL (torch.distributed backend "nccl").

A "step" is one full backbone + out_layer forward (a1-a13 of SURVEY.md section 8) over one batch of
synthetic 64-atom fcc crystals already resident in HBM (BASELINE.json configs[2]: 1000 crystals,
cutoff 5 A, 1152 edges each), per rank.  With N ranks the crystals are sharded by batch index
(rank r owns crystals [r*B, (r+1)*B)), no data-path collective, and each step ends with ONE RCCL
all_gather_into_tensor of the [B,21] predictions (configs[4]).  value = edge tensor-products/s over
all ranks: one edge-TP = one (edge, conv layer) evaluation, 4 conv layers => 4 per edge per forward.

Printed by rank 0: one JSON line with the driver contract fields plus
  "roofline"     -- the dominant kernel (tp_fused_kernel, mean over its launches) against the HBM roofline
  "cpu_baseline" -- the CPU oracle (a restatement of the reference's e3nn path, NOT e3nn itself;
                    e3nn cannot be installed on either box) timed on a bounded sample, rank 0, N=1
  "extras"       -- short driver-timed runs of the other single-GPU configurations (N = 1 only; --no-extras skips):
                    configs[1] n100 forward, configs[3] training step (batch 32, eager and hipGraph) with the roofline
                    of its dominant kernel, predict() end to end at batch_size 200 / 1000
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# torch / numpy are imported inside run_rank(): the launcher path below must stay free of anything GPU

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_ACHIEVABLE = 6.29e12  # B/s, the same guide's measured float4 copy (79 % of spec)
MFMA_F32_PEAK = 157.3e12  # flop/s, dense fp32 MFMA (same guide)
MFMA_F16_PEAK = 2.5e15    # flop/s, dense fp16/bf16 MFMA
B_ALG_PER_EDGE_TP = 4816.0  # mean algorithmic bytes per edge-TP, 2-kernel architecture


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 10 + 50 forwards = 0.3 s of device time (the first few forwards after the seconds of host-side model
    # construction run at ramping clocks: 3 warm-up steps measured 1.5 % below the steady state)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--crystals", type=int, default=1000, help="crystals per rank per step (one batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="crystals in the CPU-oracle sample batch")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] / configs[3] / predict() extras")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL path (init, barrier, all_gather) even with one rank (same as MATTEN_FORCE_DIST=1)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend of the N > 1 path: nccl (= RCCL, the real one) or gloo (REHEARSAL only: "
                         "the [B,21] predictions are gathered through host tensors)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL only: every local rank runs on cuda:0 (RCCL refuses two ranks on one device, so this "
                         "needs --backend gloo); exercises the N > 1 branch on a 1-GPU box, measures nothing")
    ap.add_argument("--no-calibration", action="store_true", help="skip the fixed VALU / copy calibration kernels")
    ap.add_argument("--no-full-layers", action="store_true",
                    help="skip the second timed loop with the FULL last conv layer (value_full_layers): the counter-collection runs "
                         "(tools/collect_traffic.sh, collect_valu.sh) must see the launches of the headline loop only")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """Start the N ranks as a child `torch.distributed.run` (this process has initialised nothing on the GPU and never
    replaces itself: a child process, not an exec) and hand its exit code back."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    if args.force_dist:
        env["MATTEN_FORCE_DIST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def _pmc_traffic(kernel: str, field: str = "hbm_bytes_mean_launch"):
    """(HBM bytes per launch of `kernel`, where that number comes from) from the committed rocprofv3 PMC run
    (profiles/hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 x2 correction on
    FETCH_SIZE), or (None, None).  'hbm_bytes_mean_launch' averages over the kernel's launches of a forward,
    'hbm_bytes_per_launch' is the largest."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            d = json.load(f)
        return float(d["kernels"][kernel][field]), (f"profiles/hbm_traffic.json (tag {d.get('tag', '?')}"
                                                    f"{', commit ' + d['commit'] if d.get('commit') else ''}): a separate "
                                                    "rocprofv3 --pmc run of bench.py --steps 2, not this run")
    except Exception:
        return None, None


def _train_traffic(kernel_prefix: str):
    """(HBM bytes per launch of the training step's kernel, source) from the committed PMC run of tools/train_b2048.py
    (profiles/train_b2048_traffic.json: tools/collect_train_traffic.sh, separate FETCH_SIZE / WRITE_SIZE passes), or (None, None)"""
    try:
        with open(os.path.join(ROOT, "profiles", "train_b2048_traffic.json")) as f:
            d = json.load(f)
        k = next(k for k in d["kernels"] if k.startswith(kernel_prefix))
        return float(d["kernels"][k]["hbm_bytes_mean_launch"]), (
            f"profiles/train_b2048_traffic.json (tag {d.get('tag', '?')}{', commit ' + d['commit'] if d.get('commit') else ''}): "
            f"a separate rocprofv3 --pmc run of tools/train_b2048.py at batch {d.get('batch')}, not this run; FETCH_SIZE x2 + WRITE_SIZE")
    except Exception:
        return None, None


def calibrate(dev, copy_floats: int = 1 << 28, valu_iters: int = 4096, reps: int = 5):
    """Fixed, model-independent work timed with HIP events on the current stream (csrc/calib.hip): what THIS box at THIS
    moment sustains.  Two lines of the same commit taken on different pool machines (or DVFS states) differ in these
    numbers the way their ms_per_step differ; a kernel change shows in ms_per_step only.
      valu: 8 waves per SIMD of dependent fp32 FMA chains on every CU -> ns per wave64 VALU instruction and SIMD, and the
            effective shader clock under that load (s_memtime ticks per 100 MHz s_memrealtime tick)
      copy: 1 GiB read + 1 GiB written, 16 bytes per lane -> GB/s"""
    import ctypes

    from matten_amd import _lib, lab, ops

    lib = lab.load()    # libmatten_lab.so (make lab): measurement helpers, not part of the product library
    stream = ops._stream()
    src = torch.empty(copy_floats, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    scratch = torch.zeros(4, dtype=torch.float32, device=dev)
    clocks = torch.zeros(2, dtype=torch.int64, device=dev)

    def timed(fn):
        fn()                                   # untimed first launch (code object load, clocks ramp)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        return ms[len(ms) // 2]

    valu_ms = timed(lambda: _lib.check(lib.matten_calib_valu(valu_iters, scratch.data_ptr(), clocks.data_ptr(), stream),
                                       "matten_calib_valu"))
    ticks, ref100 = (int(v) for v in clocks.tolist())
    copy_ms = timed(lambda: _lib.check(lib.matten_calib_copy(src.data_ptr(), dst.data_ptr(), copy_floats, stream),
                                       "matten_calib_copy"))
    insts = int(lib.matten_calib_valu_insts_per_simd(valu_iters))
    del src, dst
    return {
        "valu": {"wave_insts_per_simd": insts, "ms": valu_ms, "ns_per_wave_inst_per_simd": 1e6 * valu_ms / insts,
                 "sclk_mhz_under_valu_load": 100.0 * ticks / ref100 if ref100 else None,
                 # the first wave's own view: ticks it spent / instructions it issued, over the 8 waves that share its SIMD
                 "cycles_per_wave_inst_per_simd": ticks / float(insts) if insts else None},
        "copy": {"bytes_moved": 8 * copy_floats, "ms": copy_ms, "GBps": 8 * copy_floats / (copy_ms * 1e-3) / 1e9},
    }


def _valu_profile():
    """per-launch SQ instruction counts of tp_fused_kernel from the committed rocprofv3 PMC run (profiles/tp_fused_valu.json,
    written by tools/collect_valu.sh: separate --pmc passes of this command with --steps 2), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "tp_fused_valu.json")) as f:
            return json.load(f)
    except Exception:
        return None


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _time_cpu(fn, warmup: int = 3, n: int = 10, budget_s: float = 12.0):
    """SURVEY 8d protocol: 3 warm-up + median of 10 timed forwards; a time budget bounds slow configurations (at least
    3 timed forwards are taken).  -> (median seconds, forwards timed, last result)"""
    t_start = time.perf_counter()
    res = None
    for i in range(warmup):
        res = fn()
        if time.perf_counter() - t_start > budget_s / 2 and i >= 0:
            break
    times = []
    while len(times) < n and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
        t1 = time.perf_counter()
        res = fn()
        times.append(time.perf_counter() - t1)
    times.sort()
    return times[len(times) // 2], len(times), res


def cpu_baseline(args, model, graphs, ds, gpu_out, n_layers):
    """The CPU oracle (kind "port": the builder's restatement of the reference's e3nn path -- e3nn itself cannot be
    installed on either box) on bounded samples of configs 1-3, eval mode, no_grad, fp32, same weights as the GPU model
    for the headline configuration."""
    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
    from matten_amd.data.io import structures_from_json
    from oracle.matten_ref.model import ScalarTensorOracle

    # The oracle's per-path einsums are small: beyond ~16 threads it gets slower, not faster
    # (256 host threads on the MI355X box: >100x slower), so the baseline uses at most 16.
    nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    ref = ScalarTensorOracle(dict(PAPER_HPARAMS), ds).eval()
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=False)
    sample = collate(graphs[: args.cpu_sample])
    e_sample = int(sample["edge_index"].shape[1])
    with torch.no_grad():
        med, n_timed, want = _time_cpu(lambda: ref.decode(dict(sample)))
    err = (gpu_out[: args.cpu_sample].cpu() - want).abs().max().item()
    base = {
        "value": e_sample * n_layers / med,
        "unit": "edge-TP/s",
        "cores": nthreads,
        "cpu_model": _cpu_model(),
        "host_cores_available": os.cpu_count(),
        "kind": "port",
        "sample": f"configs[2]: {args.cpu_sample} fcc-64 crystals ({e_sample} edges) per forward, median of {n_timed} "
                  f"forwards after 3 warm-up; pure-PyTorch fp32 restatement of the e3nn path (not e3nn)",
        "crystals_per_sec": args.cpu_sample / med,
        "max_abs_diff_vs_gpu": err,
    }
    # configs[0] (Si diamond, the reference's README example) and configs[1] (the reference's n100 example set)
    others = {}
    a = 5.46
    si = crystal_graph(np.array([[0.0, 0, 0], [a / 4, a / 4, a / 4]]),
                       np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]]), np.array([14, 14]), 5.0)
    n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
    g100 = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in n100]
    for name, gs in (("configs[0] Si diamond", [si]), ("configs[1] n100", g100)):
        species = sorted({int(z) for g in gs for z in g["atomic_numbers"].tolist()})
        torch.manual_seed(35)
        r = ScalarTensorOracle(dict(PAPER_HPARAMS), {"allowed_species": species,
                                                     "average_num_neighbors": average_num_neighbors(gs)}).eval()
        b = collate(gs)
        with torch.no_grad():
            m, k, _ = _time_cpu(lambda: r.decode(dict(b)), budget_s=10.0)
        e = int(b["edge_index"].shape[1])
        others[name] = {"crystals": len(gs), "atoms": int(b["pos"].shape[0]), "edges": e, "ms_per_forward": 1e3 * m,
                        "edge_TP_per_sec": e * n_layers / m, "crystals_per_sec": len(gs) / m, "forwards_timed": k}
    base["other_configs"] = others
    return base


def extras(dev):
    """Short timed runs of the remaining single-GPU configurations of BASELINE.json (random-init weights, synthetic or
    repo-held inputs; the Zenodo training set and the checkpoint are not available): see the module docstring."""
    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd import ops
    from matten_amd import predict as P
    from matten_amd.data import synthetic
    from matten_amd.data.graph import average_num_neighbors, batch_graphs_gpu, collate, crystal_graph
    from matten_amd.data.io import structures_from_json
    from matten_amd.graphs import GraphedForward, GraphedTrainStep
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    ex = {}
    n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
    species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
    n_layers = PAPER_HPARAMS["num_layers"] + 1

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, r

    # ---- configs[1]: the reference's n100 example set as ONE batch, paper hparams, 73 species ----
    torch.manual_seed(35)
    ds = {"allowed_species": species, "average_num_neighbors": 30.4017}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
    batch = batch_graphs_gpu([(s["cart_coords"], s["lattice"], s["atomic_numbers"]) for s in n100], 5.0, dev)
    E = int(batch["edge_index"].shape[1])
    with torch.no_grad():
        t_eager, _ = timed(lambda: model(dict(batch))[0]["elastic_tensor_full"], 5, 50)
        g = GraphedForward(model, batch)
        t_graph, _ = timed(lambda: g(batch), 5, 50)
    ex["configs[1]_n100_forward"] = {
        "crystals": len(n100), "atoms": int(batch["pos"].shape[0]), "edges": E, "species": len(species), "dtype": "f32",
        "ms_per_forward_eager": 1e3 * t_eager, "ms_per_forward_hipgraph": 1e3 * t_graph,
        "edge_TP_per_sec_hipgraph": E * n_layers / t_graph, "crystals_per_sec_hipgraph": len(n100) / t_graph,
        "note": "launch-bound at this size (~45 launches): the hipGraph replay is the rate to quote",
    }
    del model, g

    # ---- configs[3]: one optimisation step, lmax = 2, 3 gated blocks, batch 32, BatchNorm batch statistics, MSE in
    # irreps space, Adam; synthetic-Zenodo-like = the n100 set's size distribution ----
    lmax2 = dict(PAPER_HPARAMS, irreps_edge_sh="0e + 1o + 2e", conv_layer_irreps="32x0o+32x0e+16x1o+16x1e+4x2o+4x2e")
    graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in n100]
    ds4 = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
    BS = 32
    tb = collate(graphs[:BS], device=dev)
    target = torch.randn(BS, 21, device=dev)

    def make():
        torch.manual_seed(3)
        m = ScalarTensorModel(backbone_hparams=dict(lmax2), dataset_hparams=ds4).to(dev).train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5, fused=True, capturable=True)

    def loss_fn(preds, t):
        return torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t)

    m_e, opt_e = make()

    def eager_step():
        loss = loss_fn(m_e(dict(tb))[0], target)
        opt_e.zero_grad()
        loss.backward()
        opt_e.step()
        return loss

    t_e, _ = timed(eager_step, 5, 20)
    ops.enable_event_timing(True)
    for _ in range(5):
        eager_step()
    torch.cuda.synchronize()
    ev = {k: sum(v) / len(v) for k, v in ops.event_timings_ms().items()}
    ops.enable_event_timing(False)
    m_g, opt_g = make()
    gs = GraphedTrainStep(m_g, opt_g, loss_fn, tb, target, warmup=3)
    t_g, _ = timed(lambda: gs.step(tb, target), 5, 30)
    # the same step with this package's flat-buffer Adam (one launch for all parameters) instead of torch's fused one
    from matten_amd.optim import FlatAdam
    torch.manual_seed(3)
    m_f = ScalarTensorModel(backbone_hparams=dict(lmax2), dataset_hparams=ds4).to(dev).train()
    gf = GraphedTrainStep(m_f, FlatAdam(m_f.parameters(), lr=1e-2, weight_decay=1e-5), loss_fn, tb, target, warmup=3)
    t_f, _ = timed(lambda: gf.step(tb, target), 5, 30)
    del m_f, gf
    # the same step with the per-edge tensors (radial weights, their gradient) stored as bf16 (fp32 arithmetic)
    from matten_amd import autograd as _ag
    _ag.set_edge_storage_dtype(torch.bfloat16)
    try:
        m_b, opt_b = make()
        gb = GraphedTrainStep(m_b, opt_b, loss_fn, tb, target, warmup=3)
        t_b, _ = timed(lambda: gb.step(tb, target), 5, 30)
        del m_b, opt_b, gb
    finally:
        _ag.set_edge_storage_dtype(torch.float32)
    # a large batch of the same set (2048 crystals: the n100 sample tiled), where the step is kernel-bound
    BL = 2048
    tbl = collate([graphs[i % len(graphs)] for i in range(BL)], device=dev)
    target_l = torch.randn(BL, 21, device=dev)
    m_l, opt_l = make()

    def large_step():
        loss = loss_fn(m_l(dict(tbl))[0], target_l)
        opt_l.zero_grad()
        loss.backward()
        opt_l.step()
        return loss

    t_l, _ = timed(large_step, 5, 10)
    El, Nl = int(tbl["edge_index"].shape[1]), int(tbl["pos"].shape[0])
    # kernel times of the large step (HIP events), for the roofline of its dominant kernel at a size that fills the GPU
    ops.enable_event_timing(True)
    for _ in range(3):
        large_step()
    torch.cuda.synchronize()
    ev_l = {k: sum(v) / len(v) for k, v in ops.event_timings_ms().items()}
    ops.enable_event_timing(False)
    convs_l = [m for m in m_l.backbone.modules() if type(m).__name__ == "PointConv"]
    convs_l = [m._view if (getattr(m, "_view", None) is not None) else m for m in convs_l]
    bw_l = []
    for c in convs_l:
        pl = c.tp.plan
        kk = next((k for k in ev_l if k.startswith("tp_backward") and k.endswith(f"d_mid={pl.d_mid}/d_in={pl.d_in}")), None)
        if kk:
            bw_l.append(((8.0 + 12.0 + 8.0 * pl.weight_numel + 4.0 * (2 * pl.d_in + pl.d_mid) / (El / Nl)) * El, ev_l[kk]))
    roof_l = None
    tr_l, tr_src = _train_traffic("tp_backward_lit_wfree_kernel")
    if bw_l:
        bm, mm = sum(b for b, _ in bw_l) / len(bw_l), sum(t for _, t in bw_l) / len(bw_l)
        roof_l = {"kernel": "tp_backward (adjoint of the uvu tensor product, matten_tp_backward_lit_wfree since round 5: w re-evaluated per "
                            "workgroup on the matrix cores, dw written; mean over the conv layers)",
                  "bound": "hbm", "achieved": bm / (mm * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                  "frac": bm / (mm * 1e-3) / HBM_PEAK, "traffic": tr_l, "traffic_source": tr_src,
                  "measured_frac": None if tr_l is None else tr_l / (mm * 1e-3) / HBM_PEAK,
                  "traffic_over_contract": None if tr_l is None else tr_l / bm,
                  "algorithmic_bytes_per_launch": bm, "avg_launch_ms": mm,
                  "contract": "ids 8 + vector 12 + w read 4 W + dw written 4 W + (x, dx rows: 2 d_in; grad rows: d_mid) / deg per "
                              "(edge, layer): the two-kernel architecture's bytes (SURVEY 8d); the w-free kernel reads 128 B of hidden "
                              "features per edge instead of the 4 W bytes of w"}
    kern_l = {k: v for k, v in ev_l.items() if k.startswith(("tp_backward", "tp_scatter", "radial_mlp", "species_linear_wgrad"))}
    del m_l, opt_l
    # the same large step replayed from a hipGraph with the flat-buffer Adam (no host time between its ~220 launches)
    torch.manual_seed(3)
    m_lg = ScalarTensorModel(backbone_hparams=dict(lmax2), dataset_hparams=ds4).to(dev).train()
    gl = GraphedTrainStep(m_lg, FlatAdam(m_lg.parameters(), lr=1e-2, weight_decay=1e-5), loss_fn, tbl, target_l, warmup=3)
    t_lg, _ = timed(lambda: gl.step(tbl, target_l), 3, 10)
    del m_lg, gl, tbl
    Et, Nt = int(tb["edge_index"].shape[1]), int(tb["pos"].shape[0])
    rec = {
        "crystals": BS, "atoms": Nt, "edges": Et, "dtype": "f32 (the reference's dtype; bf16 storage: see 'bf16')",
        "library_gemms_on_the_step": 0,
        "ms_per_step_eager": 1e3 * t_e, "ms_per_step_hipgraph": 1e3 * t_g, "crystals_per_sec_hipgraph": BS / t_g,
        "ms_per_step_hipgraph_flat_adam": 1e3 * t_f,
        "optimizer": "torch.optim.Adam(fused, capturable) for the eager / hipgraph lines; matten_amd.optim.FlatAdam (own "
                     "kernel over one flat buffer) for ms_per_step_hipgraph_flat_adam",
        "bf16": {"ms_per_step_hipgraph": 1e3 * t_b,
                 "what": "opt-in bf16 STORAGE of the per-edge tensors (radial weights w[E,W] and dL/dw), fp32 arithmetic, "
                         "node features / BatchNorm statistics / parameters fp32 (MATTEN_EDGE_STORAGE=bf16)"},
        "batch2048": {"crystals": BL, "atoms": Nl, "edges": El, "ms_per_step_eager": 1e3 * t_l, "ms_per_step_hipgraph_flat_adam": 1e3 * t_lg,
                      "crystals_per_sec": BL / min(t_l, t_lg), "dtype": "f32", "data": "the n100 sample tiled to 2048 crystals", "roofline": roof_l,
                      "kernel_ms_per_launch": kern_l},
        "data": "synthetic-Zenodo-like (first 32 crystals of the reference's n100 example, random targets)",
        "note": "with l <= 2 features the 4e output has no path: 9 of the 21 components are identically 0 (SURVEY 8d)",
        "kernel_ms_per_launch_eager": {k: v for k, v in ev.items()
                                       if k.startswith(("tp_backward", "tp_scatter", "radial_mlp"))},
    }
    # roofline of the step's dominant kernel, the tensor-product adjoint: contract bytes per (edge, layer) of the
    # two-kernel architecture = ids 8 + vector 12 + w read 4W + dw written 4W + (x and dx rows: 2 d_in, grad rows: d_mid) / deg
    convs = [m for m in m_e.backbone.modules() if type(m).__name__ == "PointConv"]
    deg = Et / Nt
    bw = []
    for c in convs:
        p = c.tp.plan
        k = next((k for k in ev if k.startswith("tp_backward") and k.endswith(f"d_mid={p.d_mid}/d_in={p.d_in}")), None)
        if k:
            bw.append(((8.0 + 12.0 + 8.0 * p.weight_numel + 4.0 * (2 * p.d_in + p.d_mid) / deg) * Et, ev[k]))
    if bw:
        bytes_mean = sum(b for b, _ in bw) / len(bw)
        ms_mean = sum(t for _, t in bw) / len(bw)
        rec["roofline"] = {"kernel": "tp_backward (adjoint of the uvu tensor product + scatter; mean over the conv layers)",
                           "bound": "hbm", "achieved": bytes_mean / (ms_mean * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "frac": bytes_mean / (ms_mean * 1e-3) / HBM_PEAK, "traffic": None,
                           "algorithmic_bytes_per_launch": bytes_mean, "avg_launch_ms": ms_mean,
                           "note": "4.5 k edges per launch: far below the size that fills 256 CUs (launch-bound regime)"}
    ex["configs[3]_training_step_batch32"] = rec
    del m_e, m_g, gs

    # ---- predict() end to end: list of 1000 fcc-64 structures in, list of [3,3,3,3] tensors out ----
    torch.manual_seed(0)
    dsf = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=dsf).to(dev).eval()
    cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
    structs = synthetic.fcc64_structures(1000)
    P.predict(structs[:8], model=model, config=cfg)
    pr = {}
    for bs in (200, 1000):
        P.predict(structs, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            outp = P.predict(structs, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        pr[f"batch_size={bs}"] = {"ms_per_1000_structures": 1e3 * dt, "crystals_per_sec": len(structs) / dt}
    assert len(outp) == len(structs)
    # a list longer than one slab (predict.PREDICT_SLAB structures): slab k + 1 is packed on the host while slab k runs
    many = structs * 4
    P.predict(many, model=model, config=cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        outp = P.predict(many, model=model, config=cfg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    pr["4000_structures"] = {"ms_per_1000_structures": 1e3 * dt / 4, "crystals_per_sec": len(many) / dt}
    del many
    # where a 1000-structure call spends its time when nothing overlaps (each stage synchronised): host packing, species
    # check, evaluate_soa = H2D + device neighbour lists + forward + to_cartesian + D2H, the result list.  predict() itself
    # packs slab k + 1 behind the forwards of slab k (first slab PREDICT_FIRST_SLAB structures, then growing 4x)
    def _t():
        torch.cuda.synchronize()
        return time.perf_counter()

    stages = []
    for _ in range(3):
        t0 = _t(); pos_, cell_, Z_, ptr_, keep_, _f = P.pack_structures(structs)
        t1 = _t(); P.check_species(model, structs, Z=Z_, ptr=ptr_, index=keep_)
        t2 = _t(); out_, _e = P.evaluate_soa(model, pos_, cell_, Z_, ptr_, 5.0, batch_size=200)
        t3 = _t(); _res = [out_[i] for i in range(len(out_))]
        t4 = _t()
        from matten_amd.data.graph import batch_graphs_gpu_soa
        t5 = _t(); _b = batch_graphs_gpu_soa(pos_, cell_, Z_, ptr_, 5.0, dev)
        t6 = _t()
        with torch.no_grad():
            _p = model(_b)[0]["elastic_tensor_full"]
        t7 = _t(); _h = _p.cpu()
        t8 = _t()
        stages.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t6 - t5, t7 - t6, t8 - t7))
    st = [1e3 * sorted(c)[1] for c in zip(*stages)]
    pr["stage_split_1000_structures_ms"] = {
        "host_pack": st[0], "species_check": st[1], "evaluate_soa_total": st[2], "result_list": st[3],
        "inside_evaluate_soa": {"h2d_plus_device_graph_build": st[4], "forward": st[5], "d2h": st[6]},
        "note": "median of 3, every stage synchronised (serial sum); predict() overlaps host_pack of later slabs and the graph "
                "build of batch k + 1 with the forward of batch k"}
    ex["predict_end_to_end_fcc64"] = dict(pr, note="host structure dicts -> device neighbour lists -> forward -> Cartesian "
                                                     "tensors on the host (PCIe and host packing inside the time)")
    # the reference's real use: small crystals (the n100 sample, 4.7 atoms each, tiled to 1000 structures) at its default
    # batch_size 200 (predict.py:155).  evaluate_soa merges user batches up to a node budget (batch_size only bounds memory:
    # a crystal's prediction does not depend on its batch mates), so 200 and 1000 run the same forwards.
    torch.manual_seed(0)
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
    small = [n100[i % len(n100)] for i in range(1000)]
    P.predict(small[:8], model=model, config=cfg)
    pr = {}
    for bs in (200, 1000):
        P.predict(small, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            outp = P.predict(small, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        pr[f"batch_size={bs}"] = {"ms_per_1000_structures": 1e3 * dt, "crystals_per_sec": len(small) / dt}
    ex["predict_end_to_end_n100_x10"] = dict(pr, atoms=sum(len(s["atomic_numbers"]) for s in small),
                                             note="1000 structures = the reference's n100 example set tiled 10x; user "
                                                  "batches are merged up to MATTEN_PREDICT_NODE_BUDGET atoms per forward")
    return ex



def _dce_clause(model) -> str:
    """what the last conv layer really runs, for config.workload (dead-output elimination)"""
    from matten_amd.nn import conv as pconv

    full = [m for m in model.backbone.modules() if type(m).__name__ == "PointConv"]
    last = full[-1]
    if last._view is None or not pconv.DEAD_PATH_ELIMINATION:
        return "; every conv layer run in full"
    return (f"; last conv layer run for its live outputs only ({last._view.tp.plan.weight_numel}/{last.tp.plan.weight_numel} "
            f"radial-weight columns: the irreps its only reader takes) -- `value` counts 4 edge-TP per edge all the same, "
            f"`value_full_layers` is the rate with the full last layer")


def algorithmic_bytes_tp_kernel(plan, deg: float) -> float:
    """Per-edge algorithmic bytes of ONE TP+scatter launch (DESIGN.md 'kernels'): edge ids (8) + edge
    vector (12) + the per-edge weights read once (4 W) + node rows amortised over the degree."""
    return 8.0 + 12.0 + 4.0 * plan.weight_numel + 4.0 * (plan.d_in + plan.d_mid) / deg


def main():
    args = parse()
    args.force_dist = args.force_dist or os.environ.get("MATTEN_FORCE_DIST") == "1"
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_rank and (args.gpus > 1 or args.force_dist):
        sys.exit(launch_ranks(args))
    run_rank(args)


def run_rank(args):
    global np, torch
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py itself (it launches its ranks) or "
                         f"torch.distributed.run with --nproc-per-node {args.gpus}")
    # one process per GPU under torch.distributed.run; --force-dist exercises the RCCL path with a
    # single rank too (init, barrier, all_gather) so it can be smoke-tested on a 1-GPU box
    distributed = world > 1 or (args.force_dist and "RANK" in os.environ)
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("--share-gpu is a rehearsal of the N > 1 branch on one device: RCCL refuses two ranks per GPU, "
                         "add --backend gloo")
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    via_host = distributed and args.backend == "gloo"   # rehearsal: predictions gathered through host tensors
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if via_host:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd import ops
    from matten_amd.data import synthetic
    from matten_amd.data.graph import collate
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    B = args.crystals
    # one GPU: the config-3 set; N > 1: rank r owns crystals [r*B, (r+1)*B) of ONE set of N*B crystals
    graphs = synthetic.fcc64_shard(rank, world, B)
    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    torch.manual_seed(35)
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
    batch = collate(graphs, device=dev)
    n_edges = int(batch["edge_index"].shape[1])
    n_nodes = int(batch["pos"].shape[0])
    gathered = torch.empty(world * B, 21, dtype=torch.float32, device="cpu" if via_host else dev) if distributed else None

    def forward_only():
        """this rank's forward WITHOUT the gather: what rank-local side work (calibration re-warm, clock probe load) runs --
        a collective there would have no peer on the other ranks"""
        with torch.no_grad():
            preds, _ = model(dict(batch))
            return preds["elastic_tensor_full"]

    def step():
        out = forward_only()
        if distributed:
            dist.all_gather_into_tensor(gathered, out.cpu() if via_host else out)
            return gathered
        return out

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # input validation pipelined: the flags of forward i are read back (pinned memory + event) when forward i + 1 is
    # enqueued instead of stalling the host inside forward i; everything is checked by the end of the timed region
    model.set_input_checks(True if os.environ.get("MATTEN_BENCH_INPUT_CHECKS") == "immediate" else "deferred")
    for _ in range(args.warmup):
        step()
    barrier()
    calibration = None
    from matten_amd import lab as _lab_probe

    if rank == 0 and not args.no_calibration and _lab_probe.load() is not None:   # (also in the --share-gpu rehearsal)
        calibration = {"before": calibrate(dev)}
        for _ in range(2):          # the calibration kernels evicted the caches the warm-up filled
            forward_only()          # (rank-local: no collective, the other ranks are waiting in the barrier below)
    barrier()
    # HIP events around the dominant kernel (roofline) and the radial-MLP kernel (mfma) only: an event pair costs a few
    # microseconds of queue time, the timed region should not pay it for every launch
    ops.enable_event_timing(True, only=("tp_scatter", "agg_linear", "radial_hidden"))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    model.finish_input_checks()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ops.event_timings_ms()
    ops.enable_event_timing(False)
    assert os.environ.get("MATTEN_BENCH_NO_CHECK") == "1" or torch.isfinite(out).all()  # the env is for ablation builds only

    if calibration is not None:
        # the shader clock the chip sustains UNDER THE FORWARD's load: a one-wave probe on a side stream watches the clocks
        # while a few more (untimed) forwards run.  The fixed kernels of calibrate() run at full clock on boxes whose
        # power-hungry tensor-product kernels clock 7 % lower: this is the number two BENCH lines are normalised by.
        from matten_amd import lab as _mlab

        lib_ = _mlab.load()
        probe_clocks = torch.zeros(2, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(dev)
        n_probe = 6
        # 100 MHz ticks of n_probe - 1 forwards; the probe kernel accepts at most 1 s (steps above 200 ms: a shorter window)
        window_ticks = min(100_000_000, max(1, int(1e8 * (n_probe - 1) * elapsed / args.steps)))
        side.wait_stream(torch.cuda.current_stream(dev))
        forward_only()                                                           # the load is up before the probe starts
        with torch.cuda.stream(side):
            probe_rc = lib_.matten_calib_clock_probe(window_ticks, probe_clocks.data_ptr(), side.cuda_stream)
        for _ in range(n_probe):
            forward_only()                                                       # rank-local load: no collective
        model.finish_input_checks()
        torch.cuda.synchronize()
        tk, rf = (int(v) for v in probe_clocks.tolist()) if probe_rc == 0 else (0, 0)   # a failed probe never kills the line
        calibration["sclk_mhz_during_forward"] = 100.0 * tk / rf if rf else None
        calibration["after"] = calibrate(dev)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if via_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_layers = PAPER_HPARAMS["num_layers"] + 1
    edge_tp_per_step = n_edges * n_layers * world
    value = edge_tp_per_step * args.steps / elapsed
    crystals_per_s = B * world * args.steps / elapsed

    result = {
        "metric": "edge_tensor_products_per_sec",
        "value": value,
        "unit": "edge-TP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "crystals_per_sec": crystals_per_s,
        "rccl_ranks": dist.get_world_size() if (distributed and not via_host) else 0,   # 0: no RCCL process group
        # what a step returned on this rank: [B, 21] alone, [world * B, 21] after the all_gather (finite: asserted above)
        "output_shape": list(out.shape),
        "backend": ("gloo (REHEARSAL: host-side gather" + (", all ranks on cuda:0" if args.share_gpu else "") + "; not a "
                    "measurement)") if via_host else ("nccl (RCCL)" if distributed else None),
        "config": {
            "workload": "configs[2]: synthetic fcc-64 crystals (64 atoms, cutoff 5 A, 1152 edges each), "
                        "paper hparams lmax=4, eval forward backbone+out_layer, one batch per step per GPU"
                        + _dce_clause(model),
            "crystals_per_gpu_per_step": B,
            "nodes_per_gpu": n_nodes,
            "edges_per_gpu": n_edges,
            "conv_layers": n_layers,
            "sharding": f"batch-index x{world}: rank r owns crystals [r*{B}, (r+1)*{B}) of one {world * B}-crystal set, one "
                        f"all_gather of [B,21] per step" if distributed else "single GPU",
            "input_checks": "species / edge_index range flags computed every step, read back one step late (pinned "
                            "memory + event), all verified inside the timed region (model.set_input_checks('deferred'))",
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, tp_fused_kernel, averaged over its launches of the timed region (one per
        # conv layer and step) -- the same average rocprofv3 --stats reports for the kernel ----
        deg = n_edges / n_nodes
        per_kernel = {k: sum(v) / len(v) for k, v in kernel_ms.items()}
        from matten_amd.nn import conv as pconv

        full_convs = [m for m in model.backbone.modules() if type(m).__name__ == "PointConv"]
        # what an inference forward launches: the last conv layer runs as its view for the irreps the head reads
        convs = [m._view if (m._view is not None and pconv.DEAD_PATH_ELIMINATION) else m for m in full_convs]

        def contract_bytes(plan, cols, d_in_part, d_mid_part):
            """SURVEY 8d per-edge algorithmic bytes of a tensor-product launch restricted to a set of input blocks:
            edge ids (8) + edge vector (12) + their radial weights read once (4 per column) + node rows over the degree"""
            return (8.0 + 12.0 + 4.0 * cols + 4.0 * (d_in_part + d_mid_part) / deg) * n_edges

        layers = []
        for m in convs:
            p = m.tp.plan
            # the row stride of the neighbour sums names the launch: component-major rows are padded (plan_agg_linear)
            km = getattr(m, "agg_plan", None) is not None and n_nodes >= pconv.AGG_KM_MIN_ROWS   # (small batches keep mul_ir rows)
            k = f"tp_scatter/d_mid={m.agg_plan.ld if km else p.d_mid}/d_in={p.d_in}"
            kern = "tp_fused_kernel"
            rec = {"kernel": kern, "d_mid": p.d_mid, "weight_numel": p.weight_numel, "ms": per_kernel.get(k),
                   "algorithmic_bytes": contract_bytes(p, p.weight_numel, p.d_in, p.d_mid)}
            rec["achieved_GBps"] = rec["algorithmic_bytes"] / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] else None
            # what THIS design has to move per launch (w[E, W] never exists): split hidden features 128 B + harmonics row
            # 128 B + source index 4 B per edge; row pointer 4 B, the input row once and the neighbour-sum row as laid
            # out (component-major, padded) per node; the layer's pre-split A fragments
            ld_out = m.agg_plan.ld if km else p.d_mid
            rec["compulsory_bytes_fused_design"] = ((128.0 + 128.0 + 4.0) * n_edges + (4.0 + 4.0 * p.d_in + 4.0 * ld_out) * n_nodes
                                                    + 2.0 * 2 * 32 * 16 * p.fused_a_tiles)
            layers.append(rec)
        avg_ms = None
        # dominant kernel = the one with the most time per forward, averaged over ITS launches of the timed region like
        # rocprofv3 --stats does: tp_fused_kernel (tensor product + neighbour sum, agg to HBM)
        by_kernel = {}
        for r in layers:
            if r["ms"]:
                by_kernel.setdefault(r["kernel"], []).append((r["ms"], r["algorithmic_bytes"], r["compulsory_bytes_fused_design"]))
        dom_name = max(by_kernel, key=lambda kn: sum(m for m, _, _ in by_kernel[kn])) if by_kernel else None
        dom = by_kernel.get(dom_name, [])
        if dom:
            bytes_per_launch = sum(b for _, b, _ in dom) / len(dom)
            compulsory = sum(c for _, _, c in dom) / len(dom)
            avg_ms = sum(m for m, _, _ in dom) / len(dom)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic, traffic_source = _pmc_traffic(dom_name)
            result["roofline"] = {
                "kernel": f"{dom_name} (last radial-MLP layer on MFMA + CG paths + neighbour sum; "
                          f"mean over its {len(dom)} launches per forward)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9),
                "traffic": traffic,
                # NOT measured in this run: PMC counters need their own rocprofv3 passes (separate FETCH_SIZE / WRITE_SIZE
                # runs of this same command, tools/collect_traffic.sh); the summary they produced is committed and read here
                "traffic_source": traffic_source,
                # `achieved` / `frac` price the CONTRACT bytes of the tensor-product share of the launch (SURVEY 8d
                # two-kernel architecture: ids, edge vector, radial weights w[E, W] read once, node rows over the degree --
                # w never exists here).  What the memory system really moves:
                "measured_GBps": None if traffic is None else traffic / (avg_ms * 1e-3) / 1e9,
                "measured_frac": None if traffic is None else traffic / (avg_ms * 1e-3) / HBM_PEAK,
                "physical_bound": "fp32 VALU issue at 3 waves/SIMD (DESIGN.md section 4; `valu` below); HBM is the bound "
                                  "of the contract figure only",
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # bytes the fused design cannot avoid (per_layer[].compulsory_bytes_fused_design) and how much more the
                # memory system moved: re-reads / partial-line writes show up here
                "compulsory_bytes_per_launch": compulsory,
                "traffic_over_compulsory": None if traffic is None else traffic / compulsory,
                "avg_launch_ms": avg_ms,
                "per_layer": layers,
            }
        result["kernel_ms_per_launch"] = per_kernel
        if calibration is not None:
            result["calibration"] = dict(calibration, what=(
                "fixed kernels (csrc/calib.hip) timed with HIP events right before and right after the timed region: "
                "VALU issue rate + effective shader clock, HBM copy rate; and sclk_mhz_during_forward = the average shader "
                "clock a one-wave probe saw while untimed forwards of this workload ran.  Pool boxes run the fixed kernels "
                "alike and the forward up to 9 % apart: compare ms_per_step x sclk_mhz_during_forward (= cycles per step) "
                "between two BENCH lines, not ms_per_step alone"))
            if calibration.get("sclk_mhz_during_forward"):
                result["calibration"]["mcycles_per_step"] = result["ms_per_step"] * calibration["sclk_mhz_during_forward"] * 1e-3
        # ---- what binds tp_fused_kernel physically: fp32 VALU issue (the contract figure above charges bytes the kernel
        # never moves).  Instruction counts per launch from the committed PMC run, issue rate from THIS run's calibration.
        if dom:
            from matten_amd.o3 import wigner_3j

            nnz = lambda l1, l2, l3: int((np.abs(wigner_3j(l1, l2, l3)) > 1e-9).sum())
            cg_flops = [2.0 * sum(pt.mul * nnz(pt.l1, pt.l2, pt.l3) for pt in m.tp.plan.paths) * n_edges
                        for m, r in zip(convs, layers) if r["kernel"] == dom_name and r["ms"]]
            cg_mean = sum(cg_flops) / len(cg_flops)
            vp = _valu_profile()
            n_simd = 256 * 4
            ns_ideal = 2.0 / 2.4                       # 2 cycles per wave64 fp32 instruction at 2.4 GHz (guide)
            ns_meas = calibration["before"]["valu"]["ns_per_wave_inst_per_simd"] if calibration else None
            valu = {
                "algorithmic_cg_flops_per_launch": cg_mean,
                "flop_count": "nnz-sparse Clebsch-Gordan contraction, 2 x mul x nnz(l1,l2,l3) per path and edge (SURVEY "
                              "Appendix B 'nnz' column: 800 / 15060 / 25948 / 26972 per edge for the full layers), layers "
                              "as launched; the radial MLP and the x*w products are not counted",
                "fp32_vector_peak_TFLOPs": MFMA_F32_PEAK / 1e12,
                "achieved_TFLOPs": cg_mean / (avg_ms * 1e-3) / 1e12,
                "frac_of_fp32_vector_peak": cg_mean / (avg_ms * 1e-3) / MFMA_F32_PEAK,
            }
            if vp is not None and dom_name in vp.get("kernels", {}):
                wi = float(vp["kernels"][dom_name]["SQ_INSTS_VALU_mean_launch"])
                valu.update({
                    "valu_wave_insts_per_launch": wi,
                    "insts_source": f"profiles/tp_fused_valu.json (tag {vp.get('tag', '?')}, commit {vp.get('commit', '?')}): "
                                    "separate rocprofv3 --pmc passes of this command, not this run",
                    "issue_frac_at_2_cycles_2p4GHz": wi / n_simd * ns_ideal * 1e-9 / (avg_ms * 1e-3),
                    "issue_frac_at_calibrated_rate": None if ns_meas is None else wi / n_simd * ns_meas * 1e-9 / (avg_ms * 1e-3),
                    "other_insts_per_launch": {k: v for k, v in vp["kernels"][dom_name].items()
                                               if k.endswith("_mean_launch") and k != "SQ_INSTS_VALU_mean_launch"},
                })
            result["roofline"]["valu"] = valu
            # ---- the physical floors of the dominant kernel, in time: what THIS design has to move at the memory system's
            # rates, and its sparse CG arithmetic at the fp32 vector peak.  `frac` above prices contract bytes the fused design
            # never moves (w[E, W]); frac_physical is the larger floor over the measured launch: how far the kernel is from ITS
            # OWN speed of light.  (All per mean launch, like `achieved`.)
            c_ms8, c_ms6 = 1e3 * compulsory / HBM_PEAK, 1e3 * compulsory / HBM_ACHIEVABLE
            cg_ms = 1e3 * cg_mean / MFMA_F32_PEAK
            result["roofline"]["physical"] = {
                "compulsory_bytes_per_launch": compulsory,
                "compulsory_ms_at_8TBps": c_ms8,
                "compulsory_ms_at_6.29TBps": c_ms6,
                "cg_fma_ms_at_157.3TF": cg_ms,
                "avg_launch_ms": avg_ms,
                "frac_physical": max(c_ms8, cg_ms) / avg_ms,
                "frac_physical_at_achievable_hbm": max(c_ms6, cg_ms) / avg_ms,
                "times_above_floor": avg_ms / max(c_ms8, cg_ms),
                "measured_hbm_frac": result["roofline"]["measured_frac"],
                "traffic_over_compulsory": result["roofline"]["traffic_over_compulsory"],
                "cg_frac_of_fp32_vector_peak": valu["frac_of_fp32_vector_peak"],
                "valu_issue_frac": valu.get("issue_frac_at_calibrated_rate") or valu.get("issue_frac_at_2_cycles_2p4GHz"),
                "cg_share_of_valu_insts": (cg_mean / 2.0 / 64.0 / valu["valu_wave_insts_per_launch"]
                                           if valu.get("valu_wave_insts_per_launch") else None),
                "note": "contract frac (`frac`) / measured HBM frac / traffic over compulsory / CG flops over the fp32 vector peak "
                        "side by side; the kernel is bound by fp32 VALU issue, not by either floor (DESIGN.md section 4)",
            }
        # ---- matrix-core use of the radial MLP (the only GEMM of the path): hidden layers nb -> 32 -> 32 in
        # radial_hidden_kernel (fp32 MFMA), last layer 32 -> W inside the tensor-product kernels (three fp16-split products) ----
        rh_key = next((k for k in ("radial_hidden_multi", "radial_hidden") if k in per_kernel), None)
        if layers and rh_key and any(r["ms"] for r in layers):
            nb = int(PAPER_HPARAMS.get("num_radial_basis", 8))
            # radial_hidden_multi evaluates the two hidden layers of ALL conv layers' MLPs in one launch
            n_mlps = n_layers if rh_key == "radial_hidden_multi" else 1
            hid_flops = 2.0 * (nb * 32 + 32 * 32) * n_edges * n_mlps
            hid_ms = per_kernel[rh_key]
            last_flops = [2.0 * 32 * r["weight_numel"] * n_edges for r in layers]
            tp_ms = sum(r["ms"] or 0.0 for r in layers) / len(layers)
            result["mfma"] = {
                "radial_hidden_kernel": {
                    "kernel": rh_key + ("_kernel (hidden layers of all %d conv layers' radial MLPs per launch)" % n_mlps
                                        if n_mlps > 1 else "_kernel"),
                    "flops_per_launch": hid_flops, "avg_launch_ms": hid_ms,
                    "achieved_TFLOPs": hid_flops / (hid_ms * 1e-3) / 1e12, "peak_TFLOPs": MFMA_F32_PEAK / 1e12,
                    "frac": hid_flops / (hid_ms * 1e-3) / MFMA_F32_PEAK, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                },
                "last_layer_in_tp_kernels": {
                    "algorithmic_flops_per_layer": sum(last_flops) / len(last_flops),
                    "issued_f16_flops_per_layer": 3.0 * sum(last_flops) / len(last_flops),
                    "note": "evaluated as hi.hi + 2^-11 (hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16 inside "
                            "tp_fused_kernel; matrix-pipe busy fraction from PMC in DESIGN.md section 4",
                    "f16_TFLOPs_over_kernel_time": 3.0 * sum(last_flops) / len(last_flops) / (tp_ms * 1e-3) / 1e12,
                    "peak_f16_TFLOPs": MFMA_F16_PEAK / 1e12,
                },
            }
        # two-kernel contract bytes per edge of one layer: TP launch (ids, vector, w read, node rows) + w written once
        two_kernel = lambda p: algorithmic_bytes_tp_kernel(p, deg) + 4.0 * p.weight_numel
        executed_share = sum(two_kernel(m.tp.plan) for m in convs) / sum(two_kernel(m.tp.plan) for m in full_convs)
        result["path_roofline"] = {
            "definition": "edge-TP/s x 4816 B (SURVEY 8d two-kernel algorithmic bytes of the FULL model) x executed_share "
                          "/ 8.0e12 B/s, per GPU; executed_share = contract bytes of the layers as launched (the last "
                          "conv layer without its dead output irreps) over those of the full layers",
            "executed_share": executed_share,
            "frac": value / world * B_ALG_PER_EDGE_TP * executed_share / HBM_PEAK,
        }
        # the whole forward against its own floors: every tensor-product launch's compulsory bytes + what the other kernels must
        # move at least (lin2 reads the neighbour sums once and writes the activated row; lin1 / sc / read-out one row in, one
        # out; the radial hidden launch writes 128 B per edge and layer; geometry 12 B in, 16 + 128 B out per edge)
        if dom:
            other = 0.0
            for m, r in zip(convs, layers):
                pl = m.tp.plan
                other += 4.0 * n_nodes * ((m.agg_plan.ld if getattr(m, "agg_plan", None) is not None else pl.d_mid) + 3 * pl.d_in)
            other += (128.0 * len(layers) + 12.0 + 16.0 + 128.0) * n_edges
            path_bytes = sum(r["compulsory_bytes_fused_design"] for r in layers) + other
            path_cg = sum(2.0 * sum(pt.mul * nnz(pt.l1, pt.l2, pt.l3) for pt in m.tp.plan.paths) * n_edges for m in convs)
            step_ms = result["ms_per_step"]
            result["path_roofline"]["physical"] = {
                "compulsory_bytes_per_step": path_bytes,
                "compulsory_ms_at_8TBps": 1e3 * path_bytes / HBM_PEAK,
                "compulsory_ms_at_6.29TBps": 1e3 * path_bytes / HBM_ACHIEVABLE,
                "cg_fma_ms_at_157.3TF": 1e3 * path_cg / MFMA_F32_PEAK,
                "ms_per_step": step_ms,
                "frac_physical": max(1e3 * path_bytes / HBM_PEAK, 1e3 * path_cg / MFMA_F32_PEAK) / step_ms,
                "times_above_floor": step_ms / max(1e3 * path_bytes / HBM_PEAK, 1e3 * path_cg / MFMA_F32_PEAK),
                "note": "layers as launched (last conv layer without its dead output irreps); the contract figure "
                        "path_roofline.frac charges 4816 B per edge-TP incl. w[E, W] written and re-read, which this design "
                        "never moves",
            }
        lastf, laste = full_convs[-1], convs[-1]
        result["dead_output_elimination"] = {
            "enabled": laste is not lastf,
            "what": "inference runs the last conv layer for the output irreps its only reader (the o3.Linear head onto "
                    "conv_to_output_hidden_irreps_out) takes; the other output irreps, the tensor-product paths that end in "
                    "them and their radial-weight columns are dead code in the reference graph "
                    "(model_factory/tfn_scalar_tensor.py:122-139).  Same model output (tests: "
                    "test_dead_output_elimination_matches_the_full_layer), same gradients (dead weights: exact zeros), parameters / checkpoints untouched; "
                    "MATTEN_DEAD_PATH_ELIMINATION=0 runs the full layer",
            "last_conv_layer": {"irreps_out_full": str(lastf.sc.irreps_out), "irreps_out_run": str(laste.sc.irreps_out),
                                "weight_columns_full": lastf.tp.plan.weight_numel, "weight_columns_run": laste.tp.plan.weight_numel,
                                "d_mid_full": lastf.tp.plan.d_mid, "d_mid_run": laste.tp.plan.d_mid},
        }
        if world == 1 and not distributed and not args.no_extras and os.environ.get("MATTEN_BENCH_NO_GRAPH") != "1":
            # the same forward as ONE hipGraph replay per step (all ~45 launches, CSR build included, captured once for this
            # batch shape): what the step costs without a host in the loop
            from matten_amd.graphs import GraphedForward

            with torch.no_grad():
                gf = GraphedForward(model, batch)
                for _ in range(3):
                    gf(batch)
                barrier()
                tg = time.perf_counter()
                for _ in range(20):
                    gf(batch)
                barrier()
                ms_graph = 1e3 * (time.perf_counter() - tg) / 20
            result["hipgraph_replay"] = {"ms_per_step": ms_graph, "value": n_edges * n_layers / (ms_graph * 1e-3), "steps": 20,
                                         "warmup": 3, "note": "matten_amd.graphs.GraphedForward on the bench workload; the "
                                         "headline `value` above is the eager loop"}
            del gf
            import gc

            gc.collect()               # the captured graph and its private memory pool go NOW, not at some collection
            torch.cuda.empty_cache()   # inside a later timed loop (a one-off ~100 ms stall in the extras otherwise)
        if laste is not lastf and world == 1 and not distributed and not args.no_full_layers:
            # the same workload with the FULL last layer, timed by the same loop at the same steps / warm-up (like for like
            # with SURVEY's 4816 B per edge-TP: path_roofline.frac_full_layers)
            pconv.DEAD_PATH_ELIMINATION = False
            try:
                for _ in range(args.warmup):
                    step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                model.finish_input_checks()
                barrier()
                ms_full = 1e3 * (time.perf_counter() - t1) / args.steps
            finally:
                pconv.DEAD_PATH_ELIMINATION = True
            result["ms_per_step_full_layers"] = ms_full
            result["value_full_layers"] = n_edges * n_layers / (ms_full * 1e-3)
            result["path_roofline"]["frac_full_layers"] = result["value_full_layers"] * B_ALG_PER_EDGE_TP / HBM_PEAK
            # the like-for-like pair at top level: every edge-TP of the count executed in full, priced with SURVEY's 4816 B
            result["frac_full_layers"] = result["path_roofline"]["frac_full_layers"]
            result["dead_output_elimination"]["without_it"] = {
                "ms_per_step": ms_full, "value": result["value_full_layers"], "steps": args.steps, "warmup": args.warmup}

        # ---- CPU baseline: the oracle on a bounded sample of the same workload ----
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, model, graphs, ds, out, n_layers)
        # ---- the other single-GPU configurations, timed by this same driver-run command ----
        if world == 1 and not distributed and not args.no_extras:
            model.set_input_checks(True)
            del batch, out
            torch.cuda.empty_cache()
            result["extras"] = extras(dev)
        print(json.dumps(result), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
