"""
``predict`` / ``evaluate``: the reference's public inference API (predict.py:117-245) on MI355X.

Same signature and semantics: one structure or a list in, one tensor or a list out in input order,
``None`` (plus a warning) for structures whose graph cannot be built, ``RuntimeError`` for species the
model was not trained on.  Differences, all forced by the environment: structures are duck-typed
(``.lattice.matrix`` / ``.cart_coords`` / ``.atomic_numbers``, or dicts with keys ``lattice`` /
``cart_coords`` / ``atomic_numbers``) so pymatgen is optional -- when it is importable and
``is_elasticity_tensor`` is set the result is wrapped in ``ElasticTensor`` exactly like the reference,
otherwise a ``numpy [3,3,3,3]`` array is returned -- and the checkpoint is a plain ``torch.save`` of
{"state_dict", "hyper_parameters"} or a Lightning ``.ckpt`` with the same two keys.
"""
import warnings
from pathlib import Path
from typing import Any, Dict, List, Sequence, Union

import numpy as np
import torch

from .data.graph import EdgelessStructures, batch_graphs_gpu, collate, crystal_graph, image_reach
from .model_factory.tfn_atomic_tensor import AtomicTensorModel
from .model_factory.tfn_scalar_tensor import ScalarTensorModel
from .parallel import sharded_apply
from .utils import CartesianTensorWrapper, yaml_load


def get_pretrained_model_dir(identifier: str) -> Path:
    p = Path(identifier)
    if p.exists() and p.is_dir():
        return p
    return Path(__file__).parent.parent / "pretrained" / identifier


def get_pretrained_config(identifier: str, config_filename: str = "config_final.yaml"):
    return yaml_load(get_pretrained_model_dir(identifier) / config_filename)


def get_pretrained_model(identifier: str, checkpoint: str = "model_final.ckpt", model_class=ScalarTensorModel,
                         device="cuda"):
    path = get_pretrained_model_dir(identifier) / checkpoint
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found. (The reference's own pretrained/20230627/model_final.ckpt is a large blob that is "
            "not shipped with the source tree; pass a directory holding model_final.ckpt + config_final.yaml.)"
        )
    from .checkpoint import filter_state_dict, load_checkpoint, plain, rebuild_tasks

    # restricted unpickler: the file's matten / e3nn / Lightning objects become inert placeholders (checkpoint.py)
    ckpt = load_checkpoint(path)
    hp = ckpt["hyper_parameters"]
    tasks = rebuild_tasks(hp.get("tasks"), path.parent)
    model = model_class(
        tasks=tasks, backbone_hparams=plain(hp["backbone_hparams"]), dataset_hparams=plain(hp["dataset_hparams"]),
        optimizer_hparams=plain(hp.get("optimizer_hparams")), lr_scheduler_hparams=plain(hp.get("lr_scheduler_hparams")),
    )
    # e3nn-internal buffers (output_mask, empty weight / bias placeholders, fx-graph Wigner-3j constants) and
    # Lightning metric state are dropped; anything else that does not match is an error
    state, missing, bad = filter_state_dict(ckpt["state_dict"], list(model.state_dict().keys()))
    if missing or bad:
        raise RuntimeError(f"checkpoint does not match the model: missing={missing} unexpected={bad}")
    model.load_state_dict(state, strict=True)
    return model.to(device).eval()


def _fields(s) -> Dict[str, np.ndarray]:
    if isinstance(s, dict):
        return {"lattice": np.asarray(s["lattice"]), "cart_coords": np.asarray(s["cart_coords"]),
                "atomic_numbers": np.asarray(s["atomic_numbers"])}
    return {"lattice": np.asarray(s.lattice.matrix), "cart_coords": np.asarray(s.cart_coords),
            "atomic_numbers": np.asarray(s.atomic_numbers)}


def check_species(model, structures: Sequence, Z=None, ptr=None, index=None):
    """Raise like the reference (predict.py:96-114) if a structure holds a species the model was not trained on.
    With the packed arrays of ``pack_structures`` the common all-supported case is one vectorised membership test."""
    supported = set(int(z) for z in model.hparams["dataset_hparams"]["allowed_species"])
    if Z is not None:
        bad = ~np.isin(Z, np.fromiter(supported, dtype=np.int64))
        if not bad.any():
            return
        k = int(np.searchsorted(ptr, np.nonzero(bad)[0][0], side="right") - 1)
        structures, first = [structures[index[k]]], index[k]
    else:
        first = 0
    for i, s in enumerate(structures, start=first):
        numbers = set(int(z) for z in _fields(s)["atomic_numbers"])
        if not numbers.issubset(supported):
            not_supported = ", ".join(str(z) for z in sorted(numbers - supported))
            raise RuntimeError(
                f"Cannot make predictions for structure {i}. It contains species {not_supported} not supported by "
                f"the model. The model were trained with species {supported}."
            )


def _pack_fast(structures: Sequence):
    """pack_structures for the common case -- every structure a dict whose three fields already are well-formed numpy arrays
    ([n, 3] float64 coordinates, [3, 3] float64 lattice, [n] integer species, n > 0, everything finite, cell not singular) --
    without a per-structure Python body: three list comprehensions, three concatenations and vectorised checks (0.4 instead
    of 2.1 ms per 1000 fcc-64 structures).  Anything else -> None, and the per-structure path decides and warns."""
    try:
        ps = [s["cart_coords"] for s in structures]
        cs = [s["lattice"] for s in structures]
        zs = [s["atomic_numbers"] for s in structures]
        if not ps or not all(type(p) is np.ndarray and p.ndim == 2 for p in ps):
            return None
        pos, cell, Z = np.concatenate(ps), np.concatenate(cs), np.concatenate(zs)
        sizes = np.fromiter(map(len, zs), dtype=np.int64, count=len(zs))
        n = len(ps)
        if (pos.dtype != np.float64 or cell.dtype != np.float64 or Z.dtype.kind not in "iu" or pos.ndim != 2 or pos.shape[1] != 3
                or cell.shape != (3 * n, 3) or Z.ndim != 1 or len(Z) != len(pos) or sizes.min() <= 0):
            return None
        if not np.array_equal(np.fromiter(map(len, ps), dtype=np.int64, count=n), sizes):
            return None
        cell = cell.reshape(n, 3, 3)
        if not (np.isfinite(pos).all() and np.isfinite(cell).all() and (np.abs(np.linalg.det(cell)) > 1e-12).all()):
            return None
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(sizes, out=ptr[1:])
        return pos, cell, Z.astype(np.int64, copy=False), ptr, list(range(n)), []
    except Exception:  # noqa: BLE001  (objects instead of dicts, ragged arrays, ...: the careful path handles and reports them)
        return None


def pack_structures(structures: Sequence, first: int = 0):
    """One pass over the structures -> flat struct-of-arrays batch (pos [N,3] f64, cell [B,3,3] f64, Z [N] i64,
    ptr [B+1]) of the usable ones, plus the indices that cannot be used (same role as the per-structure try/except of
    the reference dataset, dataset/structure_scalar_tensor.py:296-362): malformed arrays, no atoms, non-finite
    numbers, singular cell.  Everything after the attribute access is vectorised."""
    fast = _pack_fast(structures)
    if fast is not None:
        return fast
    pos_l, cell_l, z_l, keep, failed = [], [], [], [], []
    asarray, f64, i64 = np.asarray, np.float64, np.int64
    for i, s in enumerate(structures):
        try:
            if isinstance(s, dict):
                p, c, z = s["cart_coords"], s["lattice"], s["atomic_numbers"]
            else:
                p, c, z = s.cart_coords, s.lattice.matrix, s.atomic_numbers
            p, c, z = asarray(p, dtype=f64), asarray(c, dtype=f64), asarray(z, dtype=i64)
            if p.ndim != 2 or p.shape[1] != 3:
                p = p.reshape(-1, 3)
            if c.shape != (3, 3):
                c = c.reshape(3, 3)
            if z.ndim != 1:
                z = z.reshape(-1)
            if len(p) == 0 or len(p) != len(z):
                raise ValueError("malformed structure")
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"Failed converting structure {first + i}, Skip it. {e}")
            failed.append(i)
            continue
        pos_l.append(p); cell_l.append(c); z_l.append(z); keep.append(i)
    if not keep:
        raise RuntimeError("Cannot successfully convert any structures.")

    def flat(pos_l, cell_l, z_l):
        sizes = np.fromiter(map(len, z_l), dtype=np.int64, count=len(z_l))
        ptr = np.zeros(len(z_l) + 1, dtype=np.int64)
        np.cumsum(sizes, out=ptr[1:])
        return np.concatenate(pos_l), np.stack(cell_l), np.concatenate(z_l), ptr

    pos, cell, Z, ptr = flat(pos_l, cell_l, z_l)
    with np.errstate(all="ignore"):
        ok = np.isfinite(cell).all(axis=(1, 2)) & (np.abs(np.linalg.det(cell)) > 1e-12)
        ok &= np.logical_and.reduceat(np.isfinite(pos).all(axis=1), ptr[:-1])
    if not ok.all():
        for k in np.nonzero(~ok)[0]:
            warnings.warn(f"Failed converting structure {first + keep[k]}, Skip it. singular cell or non-finite coordinates")
            failed.append(keep[k])
        sel = np.nonzero(ok)[0]
        if len(sel) == 0:
            raise RuntimeError("Cannot successfully convert any structures.")
        keep = [keep[k] for k in sel]
        pos, cell, Z, ptr = flat([pos_l[k] for k in sel], [cell_l[k] for k in sel], [z_l[k] for k in sel])
    return pos, cell, Z, ptr, keep, sorted(failed)


_SIDE_STREAMS: Dict[Any, Any] = {}

# atoms one forward may hold when evaluate_soa coalesces user batches (MATTEN_PREDICT_NODE_BUDGET; ~30 KB of device buffers
# per atom at 30 neighbours: 2 GB at the default)
NODE_BUDGET = int(__import__("os").environ.get("MATTEN_PREDICT_NODE_BUDGET", "65536"))


REFERENCE_BATCH_SIZE = 200   # predict.py:155 of the reference


def effective_node_budget(batch_size: int, node_budget: int = None) -> int:
    """atoms a merged forward may hold: the caller's `node_budget` when given; otherwise NODE_BUDGET -- unless the caller
    LOWERED batch_size below the reference's default, which is how one makes a forward fit a small or shared GPU there:
    such batches are run as they are (0 = no merging)"""
    if node_budget is not None:
        return max(0, int(node_budget))
    return NODE_BUDGET if batch_size >= REFERENCE_BATCH_SIZE else 0


def coalesce_batches(ptr, batch_size: int, node_budget: int = None):
    """Crystal index ranges of the forwards evaluate_soa runs.  ``batch_size`` is the reference's knob for how many
    structures share a forward (predict.py:155, default 200); a crystal's prediction does not depend on its batch mates
    (tests: bitwise within a kernel path, ~1e-7 across the row-resident / streaming lin2 switch), so here it only bounds
    memory: consecutive user batches are merged while the atoms of the merged batch stay within ``node_budget``.  The
    reference's data set averages 4.7 atoms per crystal -- 200 of them are ~1000 atoms, a forward that is all launch
    latency (~45 launches), while 65536 atoms fill the GPU.  A single user batch larger than the budget is kept as it is."""
    node_budget = NODE_BUDGET if node_budget is None else node_budget
    B = len(ptr) - 1
    chunks, lo = [], 0
    while lo < B:
        hi = min(B, lo + batch_size)
        while hi < B and ptr[min(B, hi + batch_size)] - ptr[lo] <= node_budget:
            hi = min(B, hi + batch_size)
        chunks.append(np.arange(lo, hi))
        lo = hi
    return chunks


def _side_stream(device):
    """one persistent graph-construction stream per device (a fresh stream per call would also mean a fresh, empty
    allocator pool per call)"""
    key = torch.device(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(key)
    return _SIDE_STREAMS[key]


_CONVERTERS: Dict[str, CartesianTensorWrapper] = {}


def _converter(formula: str) -> CartesianTensorWrapper:
    """one converter per formula: its change-of-basis matrix stays on the device between calls (a fresh one uploads it
    behind the forward: a blocking 7 KB copy the host sits in for the whole forward)"""
    if formula not in _CONVERTERS:
        _CONVERTERS[formula] = CartesianTensorWrapper(formula)
    return _CONVERTERS[formula]


class _deferred_input_checks:
    """the per-forward range checks are read one batch late (and all of them before the results leave): the host keeps
    building the next batch instead of waiting for the flags of the one it has just enqueued"""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.modes = [(m, m.check_species) for m in self.model.modules() if hasattr(m, "check_species")]
        if hasattr(self.model, "set_input_checks"):
            self.model.set_input_checks("deferred")
        return self

    def __exit__(self, *exc):
        for m, mode in self.modes:
            m._pending = None
            m.check_species = mode
        return False


def evaluate_soa(model, pos, cell, Z, ptr, r_cut: float, batch_size: int = 200,
                 tensor_target_name: str = "elastic_tensor_full", tensor_target_formula: str = "ijkl=jikl=klij",
                 node_budget: int = None):
    """Batched forward straight from the flat arrays of ``pack_structures``.  Graphs are built on the device batch
    by batch on a SECOND stream: the neighbour search of batch k+1 (and its one host sync, the edge count) overlaps
    the forward of batch k, which runs on the caller's stream.
    -> (Cartesian tensors [B, 3, ...] on the host as one array, indices of crystals without any edge)."""
    model.eval()
    with _deferred_input_checks(model):
        return _soa_end(model, _soa_begin(model, pos, cell, Z, ptr, r_cut, batch_size, tensor_target_name,
                                          tensor_target_formula, node_budget))


def _soa_begin(model, pos, cell, Z, ptr, r_cut, batch_size, tensor_target_name, tensor_target_formula, node_budget=None):
    """enqueue every forward of the packed structures (inside _deferred_input_checks) -> (device tensors [B, 3, ...],
    indices of crystals without any edge); nothing here waits for the forwards"""
    from .data.graph import batch_graphs_gpu_soa

    converter = _converter(tensor_target_formula)
    device = model.device
    rank_dims = (3,) * len(tensor_target_formula.split("=")[0].replace("-", ""))
    B = len(ptr) - 1
    out = torch.full((B,) + rank_dims, float("nan"), device=device)
    edgeless = []
    main = torch.cuda.current_stream(device)
    side = _side_stream(device)

    def slice_of(ids):
        a, b = ptr[ids[0]], ptr[ids[-1] + 1]
        if np.all(np.diff(ids) == 1):
            return pos[a:b], cell[ids], Z[a:b], ptr[ids[0] : ids[-1] + 2] - a
        rows = np.concatenate([np.arange(ptr[i], ptr[i + 1]) for i in ids])  # crystals were dropped: regather
        sp = np.zeros(len(ids) + 1, dtype=np.int64)
        np.cumsum(ptr[ids + 1] - ptr[ids], out=sp[1:])
        return pos[rows], cell[ids], Z[rows], sp

    def build(ids):
        """graph batch of crystals `ids` on the side stream; crystals without edges are dropped and recorded"""
        with torch.cuda.stream(side):
            while len(ids):
                try:
                    return ids, batch_graphs_gpu_soa(*slice_of(ids), r_cut, device)
                except EdgelessStructures as e:
                    edgeless.extend(int(ids[k]) for k in e.indices)
                    ids = np.delete(ids, e.indices)
        return ids, None

    budget = effective_node_budget(batch_size, node_budget)
    try:
        _evaluate_soa_loop(model, coalesce_batches(ptr, batch_size, budget), build, main, side, out, edgeless, converter,
                           tensor_target_name, device)
    except torch.cuda.OutOfMemoryError:
        if budget == 0:
            raise
        # a merged forward did not fit (a shared or small GPU): start over with the caller's own batches
        torch.cuda.synchronize(device)
        torch.cuda.empty_cache()
        warnings.warn(f"a merged forward of up to {budget} atoms ran out of device memory: running batches of {batch_size} "
                      "structures unmerged (pass node_budget= to predict() to set the limit)")
        out.fill_(float("nan"))
        del edgeless[:]
        _evaluate_soa_loop(model, coalesce_batches(ptr, batch_size, 0), build, main, side, out, edgeless, converter,
                           tensor_target_name, device)
    return out, edgeless


def _soa_end(model, handle):
    out, edgeless = handle
    if hasattr(model, "finish_input_checks"):
        model.finish_input_checks()
    return out.cpu().numpy(), sorted(edgeless)


def _evaluate_soa_loop(model, chunks, build, main, side, out, edgeless, converter, tensor_target_name, device):
    with torch.no_grad():
        nxt = build(chunks[0]) if chunks else None
        for k in range(len(chunks)):
            ids, batch = nxt
            if batch is not None:
                main.wait_stream(side)
                for t in batch.values():
                    t.record_stream(main)  # allocated on the side stream, consumed on the caller's
                preds, _ = model(batch, task_name=tensor_target_name)
                p = preds[tensor_target_name]
                if p.dim() == 2:  # irreps -> Cartesian on the GPU
                    p = converter.to_cartesian(p)
                if len(ids) and ids[-1] - ids[0] + 1 == len(ids):
                    out[int(ids[0]) : int(ids[-1]) + 1] = p     # a plain range: no index upload (a blocking copy
                else:                                            # queued behind the forward would stall the host here)
                    out[torch.as_tensor(ids, device=device)] = p
            nxt = build(chunks[k + 1]) if k + 1 < len(chunks) else None  # overlaps the forward just enqueued


def build_graphs(structures: Sequence, r_cut: float, on_gpu: bool = False):
    """-> (graphs, failed indices): per-structure try/except like the reference dataset
    (dataset/structure_scalar_tensor.py:296-362).

    ``on_gpu``: only validate on the host and return (pos, cell, Z) triples; the neighbour search then runs on
    the device, batch by batch, inside ``evaluate`` (crystals that turn out to have no edge come back as NaN
    rows and are reported as failed by ``predict``)."""
    graphs, failed = [], []
    for i, s in enumerate(structures):
        try:
            f = _fields(s)
            if on_gpu:
                pos = np.asarray(f["cart_coords"], dtype=np.float64).reshape(-1, 3)
                cell = np.asarray(f["lattice"], dtype=np.float64).reshape(3, 3)
                Z = np.asarray(f["atomic_numbers"], dtype=np.int64).reshape(-1)
                if len(pos) == 0 or len(pos) != len(Z) or not np.all(np.isfinite(image_reach(pos, cell, r_cut))):
                    raise ValueError("malformed structure")
                graphs.append((pos, cell, Z))
                continue
            graphs.append(crystal_graph(f["cart_coords"], f["lattice"], f["atomic_numbers"], r_cut))
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"Failed converting structure {i}, Skip it. {e}")
            failed.append(i)
    if not graphs:
        raise RuntimeError("Cannot successfully convert any structures.")
    return graphs, failed


def evaluate(model, graphs: List, batch_size: int = 200,
             tensor_target_name: str = "elastic_tensor_full", tensor_target_formula: str = "ijkl=jikl=klij",
             distributed: bool = False, r_cut: float = None) -> List[torch.Tensor]:
    """Batched forward; returns one Cartesian tensor per graph (on the host).  With ``distributed`` the
    graphs are sharded by index over the ranks and gathered with one collective per call.

    ``graphs`` holds either host graph dicts (``crystal_graph``) or raw (pos, cell, Z) triples; triples are
    turned into a batch on the device (``batch_graphs_gpu``, needs ``r_cut``).  A triple without any edge
    yields a NaN tensor."""
    converter = _converter(tensor_target_formula)
    device = model.device
    rank_dims = (3,) * len(tensor_target_formula.split("=")[0].replace("-", ""))
    # what the model emits per crystal: the irreps row ([21] for the elasticity tensor) or, for
    # output_format == "cartesian", the tensor itself.  The rows are what crosses xGMI (north star: ONE gather of
    # [B, 21]); the change of basis to [3,3,3,3] (81 floats) runs after the gather, on every rank's full set.
    cartesian_out = getattr(model, "to_cartesian", None) is not None
    row_dims = rank_dims if cartesian_out else (converter._Q.shape[0],)

    def run(shard):
        outs = []
        with torch.no_grad():
            for lo in range(0, len(shard), batch_size):
                items = shard[lo : lo + batch_size]
                keep = list(range(len(items)))
                if isinstance(items[0], dict):
                    batch = collate(items, device=device)
                else:
                    try:
                        batch = batch_graphs_gpu(items, r_cut, device)
                    except EdgelessStructures as e:
                        keep = [i for i in keep if i not in set(e.indices)]
                        batch = batch_graphs_gpu([items[i] for i in keep], r_cut, device) if keep else None
                full = torch.full((len(items),) + row_dims, float("nan"), device=device)
                if batch is not None:
                    preds, _ = model(batch, task_name=tensor_target_name)
                    p = preds[tensor_target_name]
                    if len(keep) == len(items):
                        full = p.reshape((len(items),) + row_dims)
                    else:
                        full[torch.as_tensor(keep, device=device)] = p.reshape((len(keep),) + row_dims)
                outs.append(full)
        return torch.cat(outs, dim=0)

    model.eval()
    if distributed:
        rows = sharded_apply(run, graphs, row_dims, device)
    else:
        rows = run(graphs)
    preds = rows if cartesian_out else converter.to_cartesian(rows)   # NaN rows stay NaN tensors
    return list(preds.cpu())


def evaluate_atomic(model, graphs: List, batch_size: int = 200, tensor_target_name: str = "nmr_tensor",
                    tensor_target_formula: str = "ij=ji", r_cut: float = None) -> List[torch.Tensor]:
    """Per-atom tensors of all atoms of all structures, in input order, as ONE flat list -- what the reference's
    ``evaluate`` produces for an ``AtomicTensorModel`` (predict.py:117-148: ``predictions.extend(p)``)."""
    converter = _converter(tensor_target_formula)
    device = model.device
    outs = []
    model.eval()
    with torch.no_grad():
        for lo in range(0, len(graphs), batch_size):
            items = graphs[lo : lo + batch_size]
            batch = collate(items, device=device) if isinstance(items[0], dict) else batch_graphs_gpu(items, r_cut, device)
            preds, _ = model(batch, task_name=tensor_target_name)
            p = preds[tensor_target_name]
            if p.dim() == 2:
                p = converter.to_cartesian(p)
            outs.append(p)
    return list(torch.cat(outs, dim=0).cpu())


PREDICT_SLAB = int(__import__("os").environ.get("MATTEN_PREDICT_SLAB", "1024"))   # structures packed per slab
# (a small first slab with 4x growth -- device started after 0.3 ms of packing -- was measured in round 5 and dropped: every
# forward costs the host ~1 ms (graph-build read-back + ~45 launches), so three forwards instead of one gave back what the
# earlier start gained on 1000 fcc-64 structures, 8.1 ms either way, and small structures lost: 3.4 -> 5.3 ms per 1000; with the
# vectorised packer a HALF first slab for large cells only was measured again: 6.3-6.4 vs 6.3-6.6 ms per 1000 fcc-64
# structures, 5.1-5.3 either way at 4000 -- inside the run-to-run spread, not kept)


def _predict_slabs(model, structures, r_cut, batch_size, tensor_target_name, tensor_target_formula, node_budget=None):
    """-> (predictions of the usable structures in input order, sorted indices of the failed ones)"""
    n = len(structures)
    budget = effective_node_budget(batch_size, node_budget)

    def pack(lo, hi):
        try:
            pos, cell, Z, ptr, keep, failed = pack_structures(structures[lo:hi], first=lo)
        except RuntimeError:                       # not one usable structure in this slab (each one was warned about)
            return None, list(range(lo, hi))
        keep = [lo + k for k in keep]
        check_species(model, structures, Z, ptr, keep)   # (raises like the reference, naming the structure's own index)
        return (pos, cell, Z, ptr, keep), [lo + k for k in failed]

    failed, inflight = [], []
    model.eval()
    with _deferred_input_checks(model):
        lo, size = 0, PREDICT_SLAB
        nxt = pack(0, min(n, size)) if n else None
        while nxt is not None:
            cur, bad = nxt
            failed += bad
            lo = min(n, lo + size)
            if cur is not None:
                pos, cell, Z, ptr, keep = cur
                inflight.append((keep, _soa_begin(model, pos, cell, Z, ptr, r_cut, batch_size, tensor_target_name,
                                                  tensor_target_formula, node_budget)))
                # a slab should fill a forward: ~NODE_BUDGET atoms (1024 fcc-64 crystals, ~14 000 of the reference's
                # 4.7-atom ones), between PREDICT_SLAB and 16 PREDICT_SLAB structures
                per = max(1.0, len(pos) / max(1, len(keep)))
                size = int(min(16 * PREDICT_SLAB, max(PREDICT_SLAB, max(NODE_BUDGET, budget) / per)))
            nxt = pack(lo, min(n, lo + size)) if lo < n else None   # overlaps the forwards just enqueued
        predictions = []
        for keep, handle in inflight:
            tensors, edgeless = _soa_end(model, handle)
            for j in edgeless:
                warnings.warn(f"Failed converting structure {keep[j]}, Skip it. After eliminating self edges, no edges "
                              "remain in this system.")
            dropped = set(edgeless)
            failed += [keep[j] for j in edgeless]
            predictions += [tensors[j] for j in range(len(keep)) if j not in dropped]
    return predictions, sorted(set(failed))


def predict(
    structure,
    model_identifier="20230627",
    checkpoint: str = "model_final.ckpt",
    batch_size: int = 200,
    logger_level: str = "ERROR",
    is_elasticity_tensor: bool = True,
    is_atomic_tensor: bool = False,
    model: ScalarTensorModel = None,
    config: Dict[str, Any] = None,
    node_budget: int = None,
):
    """See the module docstring.  ``model`` / ``config`` let a caller reuse an already loaded model.
    ``batch_size`` (reference predict.py:155) is the memory knob it is there: consecutive batches are merged into one forward
    only while the merged batch stays within ``node_budget`` atoms (default MATTEN_PREDICT_NODE_BUDGET = 65536, ~2 GB) and
    only when batch_size is at least the reference's default of 200; ``node_budget=0`` never merges.  A merged forward
    that runs out of device memory is retried unmerged."""
    from .log import set_logger

    set_logger(logger_level)  # reference predict.py:194
    if is_atomic_tensor:  # reference predict.py:196-199
        is_elasticity_tensor = False
    single = not isinstance(structure, (list, tuple))
    structures = [structure] if single else list(structure)

    if model is None:
        model = get_pretrained_model(model_identifier, checkpoint,
                                     model_class=AtomicTensorModel if is_atomic_tensor else ScalarTensorModel)
    if config is None:
        config = get_pretrained_config(model_identifier)
    r_cut = config["data"]["r_cut"]
    if is_atomic_tensor:
        check_species(model, structures)
        # one tensor per atom; the reference returns them as one flat list over all structures and never unwraps a
        # single structure (predict.py:210-242).  A structure whose graph cannot be built fails the whole call
        # here: with per-atom outputs the reference's "None at the failed index" bookkeeping has no meaning.
        graphs, failed = build_graphs(structures, r_cut=r_cut, on_gpu=True)
        if failed:
            raise RuntimeError(f"Cannot build the graph of structures {failed}.")
        preds = evaluate_atomic(model, graphs, batch_size=batch_size,
                                tensor_target_name=config["data"]["tensor_target_name"],
                                tensor_target_formula=config["data"]["tensor_target_formula"], r_cut=r_cut)
        return [t.numpy() for t in preds]
    # fast path: the structures are packed into flat arrays slab by slab (PREDICT_SLAB structures each); while the
    # device runs the forwards of slab k the host packs slab k + 1 (a 1000-structure slab packs in 2-3 ms, its forwards
    # take 4-5 ms: the host work of all slabs but the first disappears behind the device)
    predictions, failed = _predict_slabs(model, structures, r_cut, batch_size, config["data"]["tensor_target_name"],
                                         config["data"]["tensor_target_formula"], node_budget)
    if not predictions:
        raise RuntimeError("Cannot successfully convert any structures.")
    if is_elasticity_tensor:
        try:
            from pymatgen.analysis.elasticity import ElasticTensor

            predictions = [ElasticTensor(t) for t in predictions]
        except ImportError:
            pass

    if failed:
        failed_set, it = set(failed), iter(predictions)
        out = [None if i in failed_set else next(it) for i in range(len(structures))]
        warnings.warn(
            "Cannot make predictions for the following structures. Their returned "
            f"elasticity tensor set to `None`: {sorted(failed_set)}."
        )
    else:
        out = predictions
    return out[0] if single else out
