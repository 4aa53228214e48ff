#!/usr/bin/env python3
"""One rank of the multi-rank predict test (tests/test_gpu_bench.py::test_predict_evaluate_sharded_over_two_ranks_on_the_device):
   dist_predict_worker.py <rank> <world> <port> <out dir> <n crystals>
Every rank builds the same seeded model on cuda:0 (two ranks share the one GPU of the test box: a rehearsal of the code path,
like bench.py --share-gpu), calls matten_amd.predict.evaluate(..., distributed=True) on the SAME list of raw (positions, cell,
species) triples -- reference predict.py:117-148 is the loop being sharded -- and saves what it got back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, out_dir, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)

from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic as S
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
from matten_amd.predict import evaluate

triples = [(np.asarray(s["cart_coords"], dtype=np.float64), np.asarray(s["lattice"], dtype=np.float64), np.asarray(s["atomic_numbers"]))
           for s in S.fcc64_structures(n, S.FCC_SEED + 7)]
torch.manual_seed(35)
ds = {"allowed_species": list(S.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to("cuda:0").eval()
calls = []
fwd = model.forward
model.forward = lambda batch, *a, **k: (calls.append(int(batch["ptr"].shape[0]) - 1), fwd(batch, *a, **k))[1]
tensors = evaluate(model, triples, batch_size=4, distributed=world > 1, r_cut=5.0)
torch.save({"tensors": torch.stack(tensors), "crystals_forwarded": sum(calls)}, os.path.join(out_dir, f"rank{rank}of{world}.pt"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
