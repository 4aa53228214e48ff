#!/usr/bin/env python3
"""One rank of the data-parallel training test (tests/test_gpu_bench.py::test_data_parallel_training_step_on_the_device):
   dist_train_worker.py <rank> <world> <port> <out dir>
lmax-2 model without normalisation (BatchNorm batch statistics are per rank by design: with them a sharded step is not the
full-batch step), 8 crystals of the reference's example set sharded 4 + 4 (world 2) or whole (world 1), FlatAdam, three steps of
matten_amd.parallel.DataParallelStep; saves the flat parameter buffer and the losses."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)

from common import LMAX2
from matten_amd.data.graph import average_num_neighbors, collate, crystal_graph
from matten_amd.data.io import structures_from_json
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.optim import FlatAdam
from matten_amd.parallel import DataParallelStep, shard_bounds

structs = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))[:8]
graphs = [crystal_graph(s["cart_coords"], s["lattice"], s["atomic_numbers"], 5.0) for s in structs]
species = sorted({int(z) for s in structs for z in s["atomic_numbers"]})
ds = {"allowed_species": species, "average_num_neighbors": average_num_neighbors(graphs)}
torch.manual_seed(35 + rank)          # ranks start from DIFFERENT weights: the constructor's broadcast must align them
model = ScalarTensorModel(backbone_hparams=dict(LMAX2, normalization=None), dataset_hparams=ds).to("cuda:0").train()
if world == 1:                        # the single-process reference starts from what rank 0 starts from
    torch.manual_seed(35)
    model = ScalarTensorModel(backbone_hparams=dict(LMAX2, normalization=None), dataset_hparams=ds).to("cuda:0").train()
target = torch.randn(8, 21, generator=torch.Generator().manual_seed(9)).to("cuda:0")
opt = FlatAdam(model.parameters(), lr=1e-2, weight_decay=1e-5)
dp = DataParallelStep(model, opt, lambda preds, t: torch.nn.functional.mse_loss(preds["elastic_tensor_full"], t))
lo, hi = shard_bounds(8, rank, world)
batch = collate(graphs[lo:hi], device="cuda:0")
losses = [float(dp.step(dict(batch), target[lo:hi], 8)) for _ in range(3)]
dp.sync_buffers()
torch.save({"flat": opt.flat_params.detach().cpu(), "losses": losses}, os.path.join(out_dir, f"train{rank}of{world}.pt"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
