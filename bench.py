#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X message-passing hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

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
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12  # flop/s, dense fp32 MFMA (same guide)
MFMA_F16_PEAK = 2.5e15    # flop/s, dense fp16/bf16 MFMA
B_ALG_PER_EDGE_TP = 4816.0  # mean algorithmic bytes per edge-TP, 2-kernel architecture


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--crystals", type=int, default=1000, help="crystals per rank per step (one batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="crystals in the CPU-oracle sample batch")
    return ap.parse_args()


def _pmc_traffic(kernel: str, field: str = "hbm_bytes_mean_launch"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC run (profiles/hbm_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 x2 correction on FETCH_SIZE), or None.
    'hbm_bytes_mean_launch' averages over the kernel's launches of a forward, 'hbm_bytes_per_launch' is the largest."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            return float(json.load(f)["kernels"][kernel][field])
    except Exception:
        return None


def algorithmic_bytes_tp_kernel(plan, deg: float) -> float:
    """Per-edge algorithmic bytes of ONE TP+scatter launch (DESIGN.md 'kernels'): edge ids (8) + edge
    vector (12) + the per-edge weights read once (4 W) + node rows amortised over the degree."""
    return 8.0 + 12.0 + 4.0 * plan.weight_numel + 4.0 * (plan.d_in + plan.d_mid) / deg


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # one process per GPU under torch.distributed.run; MATTEN_FORCE_DIST=1 exercises the RCCL path with a
    # single rank too (init, barrier, all_gather) so it can be smoke-tested on a 1-GPU box
    distributed = world > 1 or (os.environ.get("MATTEN_FORCE_DIST") == "1" and "RANK" in os.environ)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd import ops
    from matten_amd.data import synthetic
    from matten_amd.data.graph import collate
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    B = args.crystals
    # rank r owns crystals [r*B, (r+1)*B) of the global synthetic set (seed offset per shard)
    graphs = synthetic.fcc64_graphs(B, seed=synthetic.FCC_SEED + rank)
    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    torch.manual_seed(35)
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to(dev).eval()
    batch = collate(graphs, device=dev)
    n_edges = int(batch["edge_index"].shape[1])
    n_nodes = int(batch["pos"].shape[0])
    gathered = torch.empty(world * B, 21, dtype=torch.float32, device=dev) if distributed else None

    def step():
        with torch.no_grad():
            preds, _ = model(dict(batch))
            out = preds["elastic_tensor_full"]
            if distributed:
                dist.all_gather_into_tensor(gathered, out)
                return gathered
            return out

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ops.enable_event_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ops.event_timings_ms()
    ops.enable_event_timing(False)
    assert os.environ.get("MATTEN_BENCH_NO_CHECK") == "1" or torch.isfinite(out).all()  # the env is for ablation builds only

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
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
        "config": {
            "workload": "configs[2]: synthetic fcc-64 crystals (64 atoms, cutoff 5 A, 1152 edges each), "
                        "paper hparams lmax=4, eval forward backbone+out_layer, one batch per step per GPU",
            "crystals_per_gpu_per_step": B,
            "nodes_per_gpu": n_nodes,
            "edges_per_gpu": n_edges,
            "conv_layers": n_layers,
            "sharding": f"batch-index x{world}, one all_gather of [B,21] per step" if distributed else "single GPU",
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, tp_fused_kernel, averaged over its launches of the timed region (one per
        # conv layer and step) -- the same average rocprofv3 --stats reports for the kernel ----
        deg = n_edges / n_nodes
        per_kernel = {k: sum(v) / len(v) for k, v in kernel_ms.items()}
        tp_plans = [m.tp.plan for m in model.backbone.modules() if hasattr(m, "tp") and hasattr(m.tp, "plan")]
        tp_keys = [f"tp_scatter/d_mid={p.d_mid}" for p in tp_plans]
        if tp_plans and all(k in per_kernel for k in tp_keys):
            layer_bytes = [algorithmic_bytes_tp_kernel(p, deg) * n_edges for p in tp_plans]
            layer_ms = [per_kernel[k] for k in tp_keys]
            bytes_per_launch = sum(layer_bytes) / len(layer_bytes)
            avg_ms = sum(layer_ms) / len(layer_ms)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = _pmc_traffic("tp_fused_kernel")
            result["roofline"] = {
                "kernel": f"tp_fused_kernel (radial GEMM + CG paths + neighbour sum; mean over its {len(tp_plans)} launches "
                          "per forward, one per conv layer)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9),
                "traffic": traffic,
                # `achieved` / `frac` price the CONTRACT bytes (SURVEY 8d two-kernel architecture: they include 4 W
                # bytes per edge for a w[E,W] this kernel never materialises).  What the memory system really moves:
                "measured_GBps": None if traffic is None else traffic / (avg_ms * 1e-3) / 1e9,
                "measured_frac": None if traffic is None else traffic / (avg_ms * 1e-3) / HBM_PEAK,
                "physical_bound": "fp32 VALU issue at 3 waves/SIMD (DESIGN.md section 4); HBM is the bound of the "
                                  "contract figure only",
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_ms,
                "per_layer": [
                    {"d_mid": p.d_mid, "weight_numel": p.weight_numel, "ms": ms, "algorithmic_bytes": by,
                     "achieved_GBps": by / (ms * 1e-3) / 1e9}
                    for p, ms, by in zip(tp_plans, layer_ms, layer_bytes)
                ],
                "last_layer_traffic": _pmc_traffic("tp_fused_kernel", "hbm_bytes_per_launch"),
            }
        result["kernel_ms_per_launch"] = per_kernel
        # ---- matrix-core use of the radial MLP (the only GEMM of the path): hidden layers nb -> 32 -> 32 in
        # radial_hidden_kernel (fp32 MFMA), last layer 32 -> W inside tp_fused_kernel (three fp16-split products) ----
        if tp_plans and "radial_hidden" in per_kernel:
            nb = int(PAPER_HPARAMS.get("num_radial_basis", 8))
            hid_flops = 2.0 * (nb * 32 + 32 * 32) * n_edges
            hid_ms = per_kernel["radial_hidden"]
            last_flops = [2.0 * 32 * p.weight_numel * n_edges for p in tp_plans]
            result["mfma"] = {
                "radial_hidden_kernel": {
                    "flops_per_launch": hid_flops, "avg_launch_ms": hid_ms,
                    "achieved_TFLOPs": hid_flops / (hid_ms * 1e-3) / 1e12, "peak_TFLOPs": MFMA_F32_PEAK / 1e12,
                    "frac": hid_flops / (hid_ms * 1e-3) / MFMA_F32_PEAK, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                },
                "last_layer_in_tp_fused": {
                    "algorithmic_flops_per_launch": sum(last_flops) / len(last_flops),
                    "issued_f16_flops_per_launch": 3.0 * sum(last_flops) / len(last_flops),
                    "note": "evaluated as hi.hi + 2^-11 (hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16 inside the kernel the "
                            "HBM roofline above prices; matrix-pipe busy fraction from PMC in DESIGN.md section 4",
                    "f16_TFLOPs_over_kernel_time": 3.0 * sum(last_flops) / len(last_flops) / (avg_ms * 1e-3) / 1e12,
                    "peak_f16_TFLOPs": MFMA_F16_PEAK / 1e12,
                },
            }
        result["path_roofline"] = {
            "definition": "edge-TP/s x 4816 B (SURVEY 8d two-kernel algorithmic bytes) / 8.0e12 B/s, per GPU",
            "frac": value / world * B_ALG_PER_EDGE_TP / HBM_PEAK,
        }

        # ---- CPU baseline: the oracle on a bounded sample of the same workload ----
        if world == 1 and not args.no_cpu_baseline:
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
                ref.decode(dict(sample))  # warm-up
                times = []
                t_budget = time.perf_counter()
                while len(times) < 5 and (time.perf_counter() - t_budget) < 20.0:
                    t1 = time.perf_counter()
                    want = ref.decode(dict(sample))
                    times.append(time.perf_counter() - t1)
            times.sort()
            med = times[len(times) // 2]
            got = out[: args.cpu_sample].cpu()
            err = (got - want).abs().max().item()
            result["cpu_baseline"] = {
                "value": e_sample * n_layers / med,
                "unit": "edge-TP/s",
                "cores": nthreads,
                "kind": "port",
                "sample": f"{args.cpu_sample} fcc-64 crystals ({e_sample} edges) per forward, median of {len(times)} "
                          f"forwards after 1 warm-up; pure-PyTorch fp32 restatement of the e3nn path (not e3nn)",
                "crystals_per_sec": args.cpu_sample / med,
                "max_abs_diff_vs_gpu": err,
            }
        print(json.dumps(result), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
