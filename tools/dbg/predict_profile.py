import os, sys, time, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import PAPER
from matten_amd import predict as P
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER), dataset_hparams=ds).to("cuda:0").eval()
cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
structs = synthetic.fcc64_structures(1000)
P.predict(structs[:8], model=model, config=cfg)
for bs in (200, 1000):
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        P.predict(structs, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize(); print(f"bs={bs}: {(time.perf_counter()-t)*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
P.predict(structs, model=model, config=cfg, batch_size=200)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
