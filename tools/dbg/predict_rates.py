import os, sys, time, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import PAPER_HPARAMS
from matten_amd import predict as P
from matten_amd.data.io import structures_from_json
from matten_amd.data import synthetic
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
n100 = structures_from_json(os.path.join(ROOT, "tests", "golden", "example_crystal_elasticity_tensor_n100.json"))
species = sorted({int(z) for s in n100 for z in s["atomic_numbers"]})
cfg = {"data": {"r_cut": 5.0, "tensor_target_name": "elastic_tensor_full", "tensor_target_formula": "ijkl=jikl=klij"}}
def run(model, structs, label):
    P.predict(structs[:8], model=model, config=cfg)
    for bs in (200, 1000):
        P.predict(structs, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): P.predict(structs, model=model, config=cfg, batch_size=bs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{label} batch_size={bs}: {1e3*dt*1000/len(structs):.2f} ms per 1000 -> {len(structs)/dt:.0f} crystals/s", flush=True)
torch.manual_seed(0)
m = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams={"allowed_species": species, "average_num_neighbors": 30.4}).to("cuda:0").eval()
run(m, [n100[i % 100] for i in range(1000)], "n100x10")
m2 = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams={"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}).to("cuda:0").eval()
st = synthetic.fcc64_structures(1000)
run(m2, st, "fcc64x1000")
run(m2, st * 4, "fcc64x4000")
