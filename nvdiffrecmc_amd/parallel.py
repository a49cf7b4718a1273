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


class GradientExchange:
    """The per-iteration exchange of the reference-sized parameter set (kd + ks + normal textures at 1024^2 = 37.7 MB, the probe,
    the vertices) cut into CHUNKS, each one flat fp32 bucket with its own asynchronous all-reduce: while chunk k + 1 is
    on the wire the caller already runs the parameter update of chunk k (`for k in ex.chunks(): ex.wait(k); adam[k].step()`).

    Zero-copy on the way back: after wait(k) the parameters' .grad ARE views into the reduced bucket (no unpack pass), holding the
    weighted SUM over the ranks; `scale` (= 1 / total weight, a device scalar in the uneven case folded to a float when every rank
    renders the same number of views) is what the optimizer multiplies the gradient by (FusedAdam grad_scales).  The pack is a copy
    per parameter into a preallocated bucket -- or nothing at all for a gradient its producer already wrote into the bucket (slot()) -- so
    nothing allocates after the first iteration (HIP-graph friendly).

    groups: list of lists of parameters (one list per chunk).  local_weight: views this rank renders; equal_shards: every rank has
    the same local_weight (then no weighting traffic at all: plain sum, scale = 1 / world)."""

    def __init__(self, groups, world_size=None, group=None, local_weight=1, equal_shards=True):
        self.groups = [list(g) for g in groups]
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = (world_size or (dist.get_world_size(group) if self.active else 1))
        self.local_weight, self.equal_shards = float(local_weight), bool(equal_shards)
        self.buckets, self.handles = [], []
        for g in self.groups:
            n = sum(p.numel() for p in g)
            extra = 0 if self.equal_shards else 1                       # the weight rides in the same bucket: still one collective per chunk
            self.buckets.append(torch.zeros(n + extra, dtype=torch.float32, device=g[0].device))
        self.bytes_per_step = sum(b.numel() for b in self.buckets) * 4 if (self.active and self.world > 1) else 0

    def chunks(self):
        return range(len(self.groups))

    def slot(self, param):
        """The part of its chunk's bucket reserved for `param`, shaped like it.  A producer that writes (or scatter-adds) the gradient
        straight into this view -- and hands it to autograd as the gradient -- makes pack() a no-op for that parameter: the gradient
        is born in the buffer the collective sends."""
        for g, b in zip(self.groups, self.buckets):
            off = 0
            for p in g:
                if p is param:
                    return b[off:off + p.numel()].view_as(p)
                off += p.numel()
        raise KeyError('parameter is not part of this exchange')

    def pack(self):
        """Bring the .grad of every parameter into its chunk's bucket: a copy per parameter, nothing for a gradient that already lives
        there (slot()), zeros for a missing one."""
        for g, b in zip(self.groups, self.buckets):
            n = b.numel() - (0 if self.equal_shards else 1)
            off = 0
            for p in g:
                dst = b[off:off + p.numel()]
                off += p.numel()
                if p.grad is None:
                    dst.zero_()
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad.reshape(-1))
            if not self.equal_shards:
                b[:n].mul_(self.local_weight)
                b[n:].fill_(self.local_weight)

    def start(self, skip_single=True):
        """Launch the all-reduce of every chunk (asynchronous, in chunk order).  With one rank (and skip_single) nothing is sent."""
        self.handles = [None] * len(self.buckets)
        if not self.active or (self.world == 1 and skip_single):
            return
        for k, b in enumerate(self.buckets):
            self.handles[k] = dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, k):
        """Chunk k has arrived: point the .grad of its parameters at the reduced bucket (views, no copy).  Returns the factor the
        optimizer must apply to these gradients (float, or a 0-dim device tensor for uneven shards)."""
        h = self.handles[k] if self.handles else None
        if h is not None:
            h.wait()
        b = self.buckets[k]
        n = b.numel() - (0 if self.equal_shards else 1)
        off = 0
        for p in self.groups[k]:
            p.grad = b[off:off + p.numel()].view_as(p)
            off += p.numel()
        if not self.equal_shards:
            # uneven shards: one more pass over the bucket (rare: n_views % world != 0); b[n] = sum of the weights (or this rank's own)
            b[:n].div_(b[n].clamp(min=1.0))
        return self.grad_mult

    @property
    def grad_mult(self):
        """What the optimizer multiplies the .grad views by: 1 / world after a summing all-reduce over even shards, else 1."""
        return (1.0 / self.world) if (self.equal_shards and self.active and self.world > 1) else 1.0
