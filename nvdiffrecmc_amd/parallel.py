"""Data-parallel plumbing for the one-view-per-GPU sharding: no collective on the data
path, ONE flattened fp32 all-reduce of the shared-parameter gradients per iteration (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests).  The reference has no distributed code at all.
"""
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world_size):
    """Views owned by `rank`: contiguous blocks, one view per rank when n_views == world_size.  Uneven shards
    (n_views % world_size != 0) are allowed -- allreduce_gradients(local_weight=len(shard)) keeps the batch mean exact --
    but an empty shard (n_views < world_size) is the caller's error to handle."""
    per = (n_views + world_size - 1) // world_size
    return list(range(rank * per, min(n_views, (rank + 1) * per)))


def allreduce_gradients(params, world_size=None, group=None, average=True, local_weight=None, skip_single=True):
    """Sum (or average: the loss is a mean over the batch of views, renderutils/ops.py:494) the .grad of
    `params` across ranks through ONE flat bucket.  Parameters without a gradient contribute zeros, so
    every rank sends the same layout.

    local_weight: number of views this rank rendered.  Each rank's gradient is the gradient of ITS mean over
    local_weight views; the batch mean is sum_r(w_r * g_r) / sum_r(w_r), which differs from the plain average
    whenever shard_views deals uneven shards (n_views % world != 0).  The weight travels as one extra element
    of the same bucket, so it is still one collective.  None = equal weights (plain average)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    ws = world_size or dist.get_world_size(group)
    if ws == 1 and skip_single:      # skip_single=False: run the collective anyway (single-rank RCCL smoke test)
        return 0
    grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params]
    weighted = average and local_weight is not None
    if weighted:
        w = float(local_weight)
        grads = [g * w for g in grads] + [torch.full((1,), w, dtype=torch.float32, device=grads[0].device)]
    flat = torch.cat(grads) if len(grads) > 1 else grads[0].clone()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if weighted:
        flat = flat[:-1] / flat[-1].clamp(min=1.0)
    elif average:
        flat.div_(ws)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat.numel() * 4
