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

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
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
        convs = [m for m in model.backbone.modules() if type(m).__name__ == "PointConv"]

        def contract_bytes(plan, cols, d_in_part, d_mid_part):
            """SURVEY 8d per-edge algorithmic bytes of a tensor-product launch restricted to a set of input blocks:
            edge ids (8) + edge vector (12) + their radial weights read once (4 per column) + node rows over the degree"""
            return (8.0 + 12.0 + 4.0 * cols + 4.0 * (d_in_part + d_mid_part) / deg) * n_edges

        layers = []
        for m in convs:
            p, fp = m.tp.plan, m.fused_plan
            ent = np.asarray(p.group_entries).reshape(-1, 32)
            cols = lambda ids: float(sum(int(ent[e][2]) * len(p.group_entry_paths[e]) for e in ids))

            def blocks(ids):  # floats of the input blocks a set of entries reads
                seen = {}
                for e in ids:
                    pth = p.paths[next(iter(p.group_entry_paths[e].values()))]
                    seen[pth.i_in1] = pth.mul * (2 * pth.l1 + 1)
                return float(sum(seen.values()))

            if fp is not None:
                lk = f"tp_lin2/d_out={fp.d_out}/d_in={p.d_in}"
                hk = f"tp_scatter/d_mid={fp.d_rest}/d_in={p.d_in}"
                rec = {"d_mid": p.d_mid, "weight_numel": p.weight_numel, "light_ms": per_kernel.get(lk),
                       "heavy_ms": per_kernel.get(hk, 0.0) if fp.rest is not None else 0.0,
                       "light_bytes": contract_bytes(p, cols(fp.light_ids), blocks(fp.light_ids), p.d_mid - fp.d_rest),
                       "heavy_bytes": contract_bytes(p, cols(fp.heavy_ids), blocks(fp.heavy_ids), fp.d_rest)
                       if fp.rest is not None else 0.0}
            else:
                k = f"tp_scatter/d_mid={p.d_mid}/d_in={p.d_in}"
                rec = {"d_mid": p.d_mid, "weight_numel": p.weight_numel, "light_ms": None, "heavy_ms": per_kernel.get(k),
                       "light_bytes": 0.0, "heavy_bytes": contract_bytes(p, p.weight_numel, p.d_in, p.d_mid)}
            rec["ms"] = (rec["light_ms"] or 0.0) + (rec["heavy_ms"] or 0.0)
            rec["algorithmic_bytes"] = algorithmic_bytes_tp_kernel(p, deg) * n_edges
            rec["achieved_GBps"] = rec["algorithmic_bytes"] / (rec["ms"] * 1e-3) / 1e9 if rec["ms"] else None
            layers.append(rec)
        light = [r for r in layers if r["light_ms"]]
        avg_ms = None
        # dominant kernel, averaged over its launches of the timed region like rocprofv3 --stats does: tp_fused_kernel
        # (default), or tp_lin2_kernel when the opt-in conv-fused variant is on (MATTEN_CONV_FUSED=1)
        dom = [(r["light_ms"], r["light_bytes"]) for r in light] if light else \
              [(r["heavy_ms"], r["heavy_bytes"]) for r in layers if r["heavy_ms"]]
        dom_name = "tp_lin2_kernel" if light else "tp_fused_kernel"
        if dom:
            bytes_per_launch = sum(b for _, b in dom) / len(dom)
            avg_ms = sum(m for m, _ in dom) / len(dom)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = _pmc_traffic(dom_name)
            result["roofline"] = {
                "kernel": f"{dom_name} (last radial-MLP layer on MFMA + CG paths + neighbour sum"
                          f"{' + lin2 of the l1<=1 input blocks' if light else ''}; mean over its {len(dom)} launches per "
                          "forward, one per conv layer)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9),
                "traffic": traffic,
                # `achieved` / `frac` price the CONTRACT bytes of the launch (SURVEY 8d two-kernel architecture restricted
                # to the input blocks this kernel takes: radial weights w[E, W] and agg[N, d_mid] that never exist here).
                # What the memory system really moves:
                "measured_GBps": None if traffic is None else traffic / (avg_ms * 1e-3) / 1e9,
                "measured_frac": None if traffic is None else traffic / (avg_ms * 1e-3) / HBM_PEAK,
                "physical_bound": "fp32 VALU issue at 3 waves/SIMD (DESIGN.md section 4); HBM is the bound of the "
                                  "contract figure only",
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_ms,
                "per_layer": layers,
                "edge_work_per_layer": "heavy_ms = tp_fused_kernel; light_ms = tp_lin2_kernel (only with MATTEN_CONV_FUSED=1, "
                                       "then heavy_ms covers the l1>=2 blocks only); algorithmic_bytes / achieved_GBps = "
                                       "the whole layer's contract bytes over both",
            }
        result["kernel_ms_per_launch"] = per_kernel
        # ---- matrix-core use of the radial MLP (the only GEMM of the path): hidden layers nb -> 32 -> 32 in
        # radial_hidden_kernel (fp32 MFMA), last layer 32 -> W inside the tensor-product kernels (three fp16-split products) ----
        if layers and "radial_hidden" in per_kernel:
            nb = int(PAPER_HPARAMS.get("num_radial_basis", 8))
            hid_flops = 2.0 * (nb * 32 + 32 * 32) * n_edges
            hid_ms = per_kernel["radial_hidden"]
            last_flops = [2.0 * 32 * r["weight_numel"] * n_edges for r in layers]
            tp_ms = sum(r["ms"] for r in layers) / len(layers)
            result["mfma"] = {
                "radial_hidden_kernel": {
                    "flops_per_launch": hid_flops, "avg_launch_ms": hid_ms,
                    "achieved_TFLOPs": hid_flops / (hid_ms * 1e-3) / 1e12, "peak_TFLOPs": MFMA_F32_PEAK / 1e12,
                    "frac": hid_flops / (hid_ms * 1e-3) / MFMA_F32_PEAK, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                },
                "last_layer_in_tp_kernels": {
                    "algorithmic_flops_per_layer": sum(last_flops) / len(last_flops),
                    "issued_f16_flops_per_layer": 3.0 * sum(last_flops) / len(last_flops),
                    "note": "evaluated as hi.hi + 2^-11 (hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16 inside tp_lin2_kernel / "
                            "tp_fused_kernel; matrix-pipe busy fraction from PMC in DESIGN.md section 4",
                    "f16_TFLOPs_over_kernel_time": 3.0 * sum(last_flops) / len(last_flops) / (tp_ms * 1e-3) / 1e12,
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
