"""
Multi-GPU inference: crystals are independent, so a batch is sharded by batch index across the ranks
of one node (one process per GPU, ``torch.distributed`` backend "nccl" == RCCL over xGMI) with the
model replicated, and the only exchange is ONE all-gather of the [B_local, n_out] predictions.
The reference has no multi-GPU path; this is the batch-sharded drop-in for it.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of range(n_items) owned by `rank` (first n%world ranks get +1)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_predictions(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All ranks receive the predictions of all shards in batch-index order: [n_total, ...].

    Shards may differ by one row; they are padded to the widest shard so a single
    ``all_gather_into_tensor`` moves everything (latency-bound: 86 KB per rank at 1000 crystals).
    """
    if not dist.is_available() or not dist.is_initialized():
        assert local.shape[0] == n_total
        return local
    world = dist.get_world_size(group)
    widest = -(-n_total // world)
    tail = local.shape[1:]
    # a gloo group (CPU-only collectives: the 2-rank rehearsal on one device, a CPU-side launcher) moves device rows through
    # the host; RCCL ("nccl") gathers them where they are
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.cpu() if via_host else local
    buf = src.new_zeros((widest,) + tuple(tail))
    buf[: src.shape[0]] = src
    out = src.new_empty((world * widest,) + tuple(tail))
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    if via_host:
        out = out.to(local.device)
    pieces: List[torch.Tensor] = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        pieces.append(out[r * widest : r * widest + (hi - lo)])
    return torch.cat(pieces, dim=0)


def sharded_apply(fn, items: Sequence, n_out_dims: Tuple[int, ...], device, dtype=torch.float32, group=None):
    """Run ``fn(items[lo:hi]) -> Tensor[hi-lo, *n_out_dims]`` on this rank's shard and gather all results."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(len(items), rank, world)
    local = fn(items[lo:hi]) if hi > lo else torch.zeros((0,) + tuple(n_out_dims), device=device, dtype=dtype)
    return gather_predictions(local, len(items), group=group)


# ------------------------------------------------------------------------------------------
# Data-parallel TRAINING (optional; the reference has no multi-GPU path): one process per GPU, the batch
# sharded by crystal index, the model replicated, ONE all-reduce of the gradients per step.  With matten_amd.optim.FlatAdam
# every gradient is a view into one flat buffer (3.5 M floats = 14 MB for the paper model): the whole exchange is a single
# RCCL all_reduce over xGMI -- no bucketing, nothing to overlap it with that would outlast it (the ring moves 14 MB in
# ~0.2 ms).  BatchNorm batch statistics stay per rank (torch DDP's default); the running statistics are averaged across
# the ranks on request (sync_buffers) so that every rank checkpoints the same model.
# ------------------------------------------------------------------------------------------
def all_reduce_flat(buf: torch.Tensor, group=None, average: bool = False) -> torch.Tensor:
    """sum (or mean) of `buf` over the ranks, in place.  A gloo group moves device tensors through the host (the 2-rank
    rehearsal on one GPU); RCCL reduces them where they are."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return buf
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if average:
        buf.div_(dist.get_world_size(group))
    return buf


class DataParallelStep:
    """``step(local_batch, local_target, n_global)``: forward + loss + backward on this rank's shard, one all-reduce of the
    gradients, optimiser step.  ``loss_fn(preds, target)`` must be a MEAN over the crystals it is given (e.g. MSE): the local
    loss is weighted with n_local / n_global before the backward pass, so the summed gradients are those of the mean over the
    whole batch whatever the shard sizes.  Returns the global loss (a device tensor, identical on every rank).

    The optimiser is matten_amd.optim.FlatAdam (one flat gradient buffer = one collective) or any torch optimiser (the
    gradients are flattened into a scratch buffer for the exchange and copied back)."""

    def __init__(self, model, optimizer, loss_fn, group=None, task_name: str = "elastic_tensor_full"):
        self.model, self.optimizer, self.loss_fn, self.group, self.task_name = model, optimizer, loss_fn, group, task_name
        self._flat = getattr(optimizer, "flat_grads", None)
        self._params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        # every rank must start from the same parameters: rank 0's are broadcast once
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for t in list(model.parameters()) + list(model.buffers()):
                if t.is_cuda and dist.get_backend(group) == "gloo":
                    host = t.detach().cpu()
                    dist.broadcast(host, src=src, group=group)
                    t.data.copy_(host)
                else:
                    dist.broadcast(t.data, src=src, group=group)
            from .nn._tables import bump_weights_epoch

            bump_weights_epoch()   # (caches derived from the weights: the parameters were written through .data)

    def step(self, batch, target, n_global: int) -> torch.Tensor:
        n_local = int(target.shape[0])
        self.optimizer.zero_grad()
        preds, _ = self.model(batch, task_name=self.task_name)
        loss = self.loss_fn(preds, target) * (n_local / float(n_global))
        loss.backward()
        if self._flat is not None:
            all_reduce_flat(self._flat, self.group)
        else:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self._params]
            flat = torch.cat([g.reshape(-1) for g in grads])
            all_reduce_flat(flat, self.group)
            o = 0
            for p, g in zip(self._params, grads):
                n = g.numel()
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
        self.optimizer.step()
        total = loss.detach().clone().reshape(1)
        all_reduce_flat(total, self.group)
        return total[0]

    def sync_buffers(self) -> None:
        """average the floating-point buffers (BatchNorm running statistics) over the ranks: call before a checkpoint / eval"""
        bufs = [b for b in self.model.buffers() if b.is_floating_point()]
        if not bufs:
            return
        flat = torch.cat([b.reshape(-1).float() for b in bufs])
        all_reduce_flat(flat, self.group, average=True)
        o = 0
        for b in bufs:
            b.copy_(flat[o:o + b.numel()].view_as(b).to(b.dtype))
            o += b.numel()
        from .nn._tables import bump_weights_epoch

        bump_weights_epoch()
