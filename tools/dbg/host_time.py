import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import PAPER_HPARAMS
from matten_amd.data import synthetic
from matten_amd.data.graph import collate
from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel
B = int(os.environ.get("B", "1000"))
graphs = synthetic.fcc64_graphs(B)
ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).to("cuda:0").eval()
batch = collate(graphs, device="cuda:0")
model.set_input_checks("deferred")
def step():
    with torch.no_grad():
        return model(dict(batch))[0]["elastic_tensor_full"]
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t = time.perf_counter(); step(); ts.append(time.perf_counter() - t)
print("host enqueue time per forward (GPU idle at start): median %.2f ms, min %.2f ms" % (sorted(ts)[5] * 1e3, min(ts) * 1e3))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("50 back-to-back forwards: %.2f ms per forward" % ((time.perf_counter() - t) / 50 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(14); st.sort_stats("tottime").print_stats(30)
