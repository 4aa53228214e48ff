"""bench.py's distributed branch under the driver's eyes (pytest -m gpu on the 1-GPU box): the RCCL code path with one
rank, and the N > 1 branch rehearsed with two gloo ranks on one device INCLUDING rank 0's calibration side work (a
collective inside it would have no peer: advisor finding of round 4).  No scaling number comes out of either."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench_line(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MATTEN_FORCE_DIST"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a CHILD process: this pytest process has initialised the GPU and must never exec
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rccl_branch_with_one_rank():
    """`bench.py --gpus 1 --force-dist`: init_process_group("nccl") = RCCL, barrier, all_gather_into_tensor of the [B, 21]
    predictions, max-over-ranks all_reduce -- everything `--gpus 8` runs, with one rank.  Rank 0's calibration (fixed
    kernels + clock probe beside untimed forwards) stays on: it must not issue a collective."""
    d = _bench_line(["--gpus", "1", "--force-dist", "--no-extras", "--no-cpu-baseline", "--steps", "3", "--warmup", "2"])
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["backend"] == "nccl (RCCL)"
    assert d["output_shape"] == [1000, 21] and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["sharding"].startswith("batch-index x1")
    assert d["calibration"]["sclk_mhz_during_forward"] and "roofline" in d


def test_bench_two_rank_rehearsal_keeps_calibration():
    """`--gpus 2 --backend gloo --share-gpu`: two ranks on cuda:0, predictions gathered through host tensors.  Measures
    nothing; what it proves is that the N > 1 branch terminates with calibration ON (rank 0 alone runs side work between
    the barriers) and gathers [2 B, 21]."""
    d = _bench_line(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--no-extras", "--no-cpu-baseline", "--steps", "3",
                     "--warmup", "1", "--crystals", "100"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and "REHEARSAL" in d["backend"]
    assert d["output_shape"] == [200, 21]
    assert d["calibration"]["before"]["valu"]["ms"] > 0 and d["calibration"]["after"]["copy"]["GBps"] > 0


def test_eight_rank_shards_tile_one_set():
    """configs[4]: rank r of 8 owns crystals [r B, (r + 1) B) of ONE set of 8 B crystals (synthetic.fcc64_shard): the shards
    laid end to end ARE that set -- positions, species, edge lists -- and the device forward of shard r equals rows
    [r B, (r + 1) B) of the forward of the whole set (a crystal's prediction does not depend on its batch mates: bitwise
    within one kernel path, to 2e-6 of the block's scale across the batch-size dependent piece lengths of small batches),
    which is what the single all_gather over xGMI relies on."""
    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd.data import synthetic as S
    from matten_amd.data.graph import collate
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    B, W = 3, 8
    whole = S.fcc64_graphs(W * B, S.FCC_SEED + 1)
    shards = [S.fcc64_shard(r, W, B) for r in range(W)]
    flat = [g for sh in shards for g in sh]
    assert len(flat) == len(whole)
    for a, b in zip(whole, flat):
        assert all(torch.equal(a[k], b[k]) for k in ("pos", "edge_index", "atomic_numbers", "cell", "edge_cell_shift"))
    torch.manual_seed(35)
    ds = {"allowed_species": list(S.FCC_METALS), "average_num_neighbors": 18.0}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to("cuda:0").eval()
    with torch.no_grad():
        full = model(collate(whole, device="cuda:0"))[0]["elastic_tensor_full"]
        for r in (0, 3, 7):
            part = model(collate(shards[r], device="cuda:0"))[0]["elastic_tensor_full"]
            want = full[r * B:(r + 1) * B]
            for lo, hi in ((0, 2), (2, 12), (12, 21)):     # per irrep block of the 21 components, against the block's scale
                err = (part[:, lo:hi] - want[:, lo:hi]).abs().max().item()
                assert err <= 2e-6 * full[:, lo:hi].abs().max().item() + 1e-12, (r, lo, err)


def test_predict_evaluate_sharded_over_two_ranks_on_the_device(tmp_path):
    """matten_amd.predict.evaluate(distributed=True) itself under two ranks (round-5 verdict: until now only a CPU stand-in went
    through sharded_apply with more than one rank): both ranks run the device forward on THEIR contiguous shard of 11 crystals
    (6 + 5: the uneven split pads the gather) -- raw triples, device neighbour lists, batches of 4 -- and every rank gets all 11
    Cartesian tensors back in input order, equal to a single process's.  gloo group, both ranks on the box's one GPU, the
    [B, 21] rows cross through host memory (parallel.gather_predictions); on a node the same call gathers over RCCL."""
    n = 11
    worker = os.path.join(ROOT, "tests", "dist_predict_worker.py")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = str(29500 + os.getpid() % 2000)
    single = subprocess.run([sys.executable, worker, "0", "1", port, str(tmp_path), str(n)], env=env, capture_output=True,
                            text=True, timeout=600, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path), str(n)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    want = torch.load(os.path.join(tmp_path, "rank0of1.pt"))
    assert want["crystals_forwarded"] == n and want["tensors"].shape == (n, 3, 3, 3, 3)
    scale = want["tensors"].abs().max().item()
    got = [torch.load(os.path.join(tmp_path, f"rank{r}of2.pt")) for r in range(2)]
    assert [g["crystals_forwarded"] for g in got] == [6, 5]          # each rank forwarded its own shard only
    for g in got:
        assert g["tensors"].shape == want["tensors"].shape and torch.isfinite(g["tensors"]).all()
        assert (g["tensors"] - want["tensors"]).abs().max().item() <= 1e-5 * scale
    assert torch.equal(got[0]["tensors"], got[1]["tensors"])


def test_data_parallel_training_step_on_the_device(tmp_path):
    """matten_amd.parallel.DataParallelStep with the real model on the device: two ranks (child processes, gloo group, both on the
    box's one GPU; on a node the same code all-reduces over RCCL), 8 crystals sharded 4 + 4, FlatAdam -- its flat gradient buffer
    is the ONE all-reduce of a step.  The ranks start from different weights (aligned by the constructor's broadcast), end on
    identical parameters bit for bit, and follow the single-process full-batch run (loss per step, parameters after three Adam
    steps) to rounding."""
    worker = os.path.join(ROOT, "tests", "dist_train_worker.py")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = str(29500 + (os.getpid() + 313) % 2000)
    single = subprocess.run([sys.executable, worker, "0", "1", port, str(tmp_path)], env=env, capture_output=True, text=True,
                            timeout=600, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    want = torch.load(os.path.join(tmp_path, "train0of1.pt"))
    got = [torch.load(os.path.join(tmp_path, f"train{r}of2.pt")) for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"]) and got[0]["losses"] == got[1]["losses"]
    assert all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(got[0]["losses"], want["losses"])), (got[0]["losses"], want["losses"])
    assert want["losses"][-1] < want["losses"][0]
    scale = want["flat"].abs().max().item()
    # three Adam steps amplify rounding differences of tiny gradients (g / sqrt(v)): parameters to 1e-3 of the largest weight
    assert (got[0]["flat"] - want["flat"]).abs().max().item() <= 1e-3 * scale
