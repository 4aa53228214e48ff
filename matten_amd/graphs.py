"""
hipGraph capture of a whole optimisation step (forward, loss, backward, optimiser) for FIXED batch shapes.

At the reference's training batch size (32 crystals, ~150 atoms, ~4.5 k edges: pretrained/20230627/config_final.yaml:3-6)
a step is ~350 kernel launches for under 3 ms of kernel time: the host cannot issue them fast enough.  Captured once
(``torch.cuda.CUDAGraph`` = hipStreamBeginCapture / hipGraphLaunch on ROCm), a step is one launch.  The kernels of
``libmatten_hip.so`` take their stream as an argument and never synchronise, so they capture as they are.

Constraints (checked): every tensor of the batch keeps its shape and dtype between steps (pad or bucket batches
upstream), the optimiser is capturable (``torch.optim.Adam(..., capturable=True)``), and the one host synchronisation
of the forward -- the species / edge_index range check of ``SpeciesEmbedding`` -- is done once, eagerly, before the
capture and switched off inside it.  Replayed batches are therefore trusted by default (the kernels clamp bad ids:
memory-safe, but the result of a malformed batch is meaningless); pass ``validate=True`` to ``step`` / ``__call__`` to
read the flag word the captured forward rewrites on every replay (one host sync after the replay) and get the same
exception the eager path raises.  A batch that carries its own CSR (the device graph builder's `_amd_perm` / `_amd_rowptr` /
`_amd_src`) is trusted beyond that: the CSR is not re-derived from edge_index, so `validate=True` vouches for the species
ids only; every replayed batch must hold exactly the tensors of the construction batch (checked, ValueError).

Construction runs `warmup` real optimisation steps on the construction batch (allocator pools, lazily built tables,
optimiser state).  Model parameters, buffers (BatchNorm running statistics) and the optimiser state are snapshotted
before and restored IN PLACE afterwards, so the first ``step()`` starts from exactly the state the caller handed over.
"""
from typing import Callable, Dict

import torch

from .nn._tables import bump_weights_epoch


# rocPRIM's radix sort switches algorithm above 2^20 keys and that path does not survive a capture on this ROCm (memory
# aperture violation at replay; eager is fine -- most likely its hipMemsetAsync nodes: a memset captured by
# matten_csr_build was not re-executed on replay either, and is a kernel now).  Sparse graphs (E <= 64 N, every
# cutoff-radius graph of this domain) are built by counting without a device-wide sort and the species grouping never
# sorts, so only DENSE graphs above that size are refused.
MAX_CAPTURED_SORT_KEYS = 1_000_000


def _check_capturable(batch) -> None:
    from . import _lib

    n_edges = int(batch["edge_index"].shape[1])
    n_nodes = int(batch["pos"].shape[0])
    sorts = n_edges > _lib.load().matten_csr_counting_max_avg_degree() * n_nodes
    if sorts and n_edges > MAX_CAPTURED_SORT_KEYS:
        raise ValueError(f"{n_edges} edges on {n_nodes} nodes: a graph this dense is built with a radix sort, which "
                         f"cannot be captured above {MAX_CAPTURED_SORT_KEYS} edges on this ROCm; run it eagerly")


def _raise_for_flags(embeds, batch) -> None:
    n_nodes = int(batch["pos"].shape[0]) if "pos" in batch else None
    for m in embeds:
        if hasattr(m, "raise_for_last_flags"):
            m.raise_for_last_flags(n_nodes)


def _copy_batch(static: Dict, batch: Dict, what: str) -> None:
    """copy a replayed batch into the captured buffers.  The batch must hold EXACTLY the tensors the capture saw: a batch of
    the device graph builder carries its destination-sorted CSR (_amd_perm / _amd_rowptr / _amd_src), which the captured
    forward then reads instead of rebuilding it -- a replayed batch without those keys would silently run on the
    construction batch's CSR against its own edge_index (and one with extra keys has tensors the graph never reads)."""
    have = {k for k, v in batch.items() if isinstance(v, torch.Tensor)}
    want = {k for k, v in static.items() if isinstance(v, torch.Tensor)}
    if have != want:
        raise ValueError(f"the captured {what} was built on a batch with tensors {sorted(want)}; this batch "
                         f"{'lacks ' + str(sorted(want - have)) if want - have else ''}"
                         f"{' and ' if (want - have and have - want) else ''}"
                         f"{'adds ' + str(sorted(have - want)) if have - want else ''}: replay batches of the same builder "
                         f"(after editing edge_index, drop the _amd_* keys from the CONSTRUCTION batch too and capture again)")
    for k in want:
        s, v = static[k], batch[k]
        if s.shape != v.shape or s.dtype != v.dtype:
            raise ValueError(f"batch['{k}'] is {tuple(v.shape)} {v.dtype}, the captured {what} takes "
                             f"{tuple(s.shape)} {s.dtype}: capture one per batch shape")
        if s.data_ptr() != v.data_ptr():
            s.copy_(v, non_blocking=True)


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn: Callable, batch: Dict[str, torch.Tensor], target: torch.Tensor,
                 warmup: int = 3, task_name: str = "elastic_tensor_full"):
        _check_capturable(batch)
        self.model, self.optimizer, self.loss_fn, self.task_name = model, optimizer, loss_fn, task_name
        dev = target.device
        self._static = {k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        self._target = target.clone()
        embeds = [m for m in model.modules() if hasattr(m, "check_species")]
        self._embeds = embeds
        # the warm-up steps below are real optimiser steps: snapshot what they mutate
        with torch.no_grad():
            snap_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
            snap_opt = {p: {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                        for p, st in optimizer.state.items()}
        # warm-up on a side stream (allocator pools, lazily built tables), with the range checks on
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
            # restore in place: the capture below must see the tensors the warm-up allocated (optimiser moments, step)
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(snap_model[k])
                for p, st in optimizer.state.items():
                    old = snap_opt.get(p)
                    for k, v in st.items():
                        if isinstance(v, torch.Tensor):
                            if old is not None and isinstance(old.get(k), torch.Tensor):
                                v.copy_(old[k])
                            else:
                                v.zero_()   # a freshly initialised Adam state: step 0, zero moments
                        elif old is not None and k in old:
                            st[k] = old[k]
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._flags = [(m, m.check_species) for m in embeds]
        for m in embeds:
            m.check_species = False   # validated above; a host sync cannot be captured
        try:
            self.graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self._loss = self._eager_step()
        finally:
            for m, f in self._flags:
                m.check_species = f

    def _eager_step(self):
        preds, _ = self.model(dict(self._static), task_name=self.task_name)
        loss = self.loss_fn(preds, self._target)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss

    def step(self, batch: Dict[str, torch.Tensor], target: torch.Tensor, validate: bool = False) -> torch.Tensor:
        """copy the batch into the captured buffers (shapes must match) and replay; returns the captured loss tensor.
        validate=True: afterwards read the replayed forward's species / edge_index flags (a host sync) and raise like
        the eager path -- the parameters have already been updated with the malformed batch by then."""
        _copy_batch(self._static, batch, "step")
        if target.data_ptr() != self._target.data_ptr():
            self._target.copy_(target, non_blocking=True)
        self.graph.replay()
        bump_weights_epoch()   # the replayed optimiser step (and BatchNorm's running statistics) moved no _version counter
        if validate:
            _raise_for_flags(self._embeds, self._static)
        return self._loss


class GraphedForward:
    """Inference forward of FIXED batch shapes as one hipGraph launch: ``out = g(batch)`` -> [B, 21] (irreps) or the
    Cartesian tensors, whatever ``model(batch)`` returns for `task_name`.  Pays where the forward is launch-bound (small
    batches: ~45 launches); at 1000 crystals per batch the GPU is busy either way."""

    def __init__(self, model, batch: Dict[str, torch.Tensor], warmup: int = 3, task_name: str = "elastic_tensor_full"):
        _check_capturable(batch)
        self.model, self.task_name = model, task_name
        dev = next(model.parameters()).device
        self._static = {k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        embeds = [m for m in model.modules() if hasattr(m, "check_species")]
        self._embeds = embeds
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                model(dict(self._static), task_name=task_name)   # range checks on: validates the captured batch
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        flags = [(m, m.check_species) for m in embeds]
        for m in embeds:
            m.check_species = False
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self._out = model(dict(self._static), task_name=task_name)[0][task_name]
        finally:
            for m, f in flags:
                m.check_species = f

    def __call__(self, batch: Dict[str, torch.Tensor], validate: bool = False) -> torch.Tensor:
        """NOTE: the species / edge_index range checks ran on the batch given at construction only; a replayed batch is
        trusted (same shapes enforced) unless validate=True (one host sync after the replay, same exceptions as eager)."""
        _copy_batch(self._static, batch, "forward")
        self.graph.replay()
        if validate:
            _raise_for_flags(self._embeds, self._static)
        return self._out
