import os, sys, time, torch, numpy as np, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
structs = synthetic.fcc64_structures(1000)
pos, cell, Z, ptr, keep, failed = P.pack_structures(structs)
for _ in range(2): P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t=time.perf_counter()
P.evaluate_soa(model, pos, cell, Z, ptr, 5.0, batch_size=200)
torch.cuda.synchronize(); dt=time.perf_counter()-t
pr.disable()
print("evaluate_soa bs=200: %.2f ms" % (dt*1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
