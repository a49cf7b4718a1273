"""Data-parallel plumbing for the one-view-per-GPU sharding: no collective on the data path, the gradients of the
shared parameters all-reduced once per iteration (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  The
reference has no distributed code at all.

GradientExchange cuts the parameter set into CHUNKS ordered by when the NEXT iteration needs them:

    chunk 0  light probe (+ vertex positions)   dense, 0.8-4.9 MB   needed first: update_pdf, BVH build, vertex frames, G-buffer
    chunk 1  kd / ks / normal textures          25-37.7 MB dense, or TILE-SPARSE: only the 768-byte tiles some rank's pixels touched

so that the caller can start the next iteration's geometry stage while the texture chunk is still on the wire and wait for it
only in front of the texture lookup (trainer.DirectLightingStep pipelines exactly that).  The tile-sparse mode sends the same
addends through the same SUM collective -- untouched tiles are zero on every rank and stay home -- so its sums are the dense
exchange's sums; it costs one small MAX all-reduce of the tile flags and one host read of the union's size.
"""
import torch
import torch.distributed as dist

TILE_FLOATS = 192        # 64 texels x 3 channels = 768 contiguous bytes: the tile of the sparse Adam path (csrc/optim.hip)


def shard_views(n_views, rank, world_size):
    """Views owned by `rank`: contiguous blocks, one view per rank when n_views == world_size.  Uneven shards
    (n_views % world_size != 0) are allowed -- GradientExchange(local_weight=len(shard), equal_shards=False) keeps the batch mean exact --
    but an empty shard (n_views < world_size) is the caller's error to handle."""
    per = (n_views + world_size - 1) // world_size
    return list(range(rank * per, min(n_views, (rank + 1) * per)))


def allreduce_gradients(params, world_size=None, group=None, average=True, local_weight=None, skip_single=True):
    """Sum (or average: the loss is a mean over the batch of views, renderutils/ops.py:494) the .grad of `params` across ranks
    through ONE flat bucket: a one-chunk GradientExchange, written back into the parameters' own .grad tensors.  Parameters
    without a gradient contribute zeros, so every rank sends the same layout.

    local_weight: number of views this rank rendered.  Each rank's gradient is the gradient of ITS mean over local_weight views; the
    batch mean is sum_r(w_r * g_r) / sum_r(w_r), which differs from the plain average whenever shard_views deals uneven shards; the
    weight travels as one extra element of the same bucket, so it is still one collective.  None = equal weights (plain average).
    Returns the bytes put into the collective (0 when nothing was sent)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    ws = world_size or dist.get_world_size(group)
    if ws == 1 and skip_single:      # skip_single=False: run the collective anyway (single-rank RCCL smoke test)
        return 0
    params = list(params)
    weighted = average and local_weight is not None
    ex = GradientExchange([params], ws, group=group, local_weight=(local_weight if weighted else 1), equal_shards=not weighted)
    ex.active = True
    own = [p.grad for p in params]
    ex.pack()
    ex.start(skip_single=False)
    f = ex.wait(0)
    if not average:
        f = 1.0
    for p, g in zip(params, own):       # the caller's tensors keep their identity: copy out of the bucket view
        red = p.grad if f == 1.0 else p.grad * f
        if g is None:
            p.grad = red.clone()
        else:
            g.copy_(red)
            p.grad = g
    return sum(p.numel() for p in params) * 4          # the payload (a weighted bucket carries one more element)


class _TileOps:
    """flags / plan / gather / scatter of the tile-sparse exchange: the HIP kernels of csrc/exchange.hip for GPU tensors; plain torch
    indexing for CPU tensors (the gloo tests of this file's logic -- the trainer never holds CPU tensors)."""

    @staticmethod
    def flags(bucket, n_tiles, tile_floats, out):
        if bucket.is_cuda:
            from . import _lib
            _lib.check(_lib.load().nvdr_tile_flags(bucket.data_ptr(), n_tiles, tile_floats, out.data_ptr(), _lib.stream_ptr()), 'nvdr_tile_flags')
        else:
            out.copy_((bucket[:n_tiles * tile_floats].view(n_tiles, tile_floats) != 0).any(1).to(torch.uint8))

    @staticmethod
    def plan(flags, n_tiles, tile_list, count):
        if flags.is_cuda:
            from . import _lib
            _lib.check(_lib.load().nvdr_tile_plan(flags.data_ptr(), n_tiles, tile_list.data_ptr(), count.data_ptr(), _lib.stream_ptr()), 'nvdr_tile_plan')
        else:
            idx = flags.nonzero().view(-1).to(torch.int32)
            tile_list[:idx.numel()] = idx
            count.fill_(idx.numel())

    @staticmethod
    def move(dense, compact, tile_list, count, n_tiles, tile_floats, gather):
        if dense.is_cuda:
            from . import _lib
            lib = _lib.load()
            if gather:
                _lib.check(lib.nvdr_tile_gather(dense.data_ptr(), tile_list.data_ptr(), count.data_ptr(), n_tiles, tile_floats, compact.data_ptr(), _lib.stream_ptr()), 'nvdr_tile_gather')
            else:
                _lib.check(lib.nvdr_tile_scatter(compact.data_ptr(), tile_list.data_ptr(), count.data_ptr(), n_tiles, tile_floats, dense.data_ptr(), _lib.stream_ptr()), 'nvdr_tile_scatter')
        else:
            k = int(count.item())
            idx = tile_list[:k].long()
            d, c = dense[:n_tiles * tile_floats].view(n_tiles, tile_floats), compact[:k * tile_floats].view(k, tile_floats)
            if gather:
                c.copy_(d.index_select(0, idx))
            else:
                d.index_copy_(0, idx, c)


class GradientExchange:
    """The per-iteration exchange of the reference-sized parameter set (kd + ks + normal textures at 1024^2 = 37.7 MB, the probe,
    the vertices) cut into CHUNKS, each one flat fp32 bucket with its own asynchronous all-reduce: while chunk k + 1 is
    on the wire the caller already runs the parameter update of chunk k (`for k in ex.chunks(): ex.wait(k); adam[k].step()`).

    Zero-copy on the way back: after wait(k) the parameters' .grad ARE views into the reduced bucket (no unpack pass), holding the
    weighted SUM over the ranks; `grad_mult` (= 1 / world over even shards) is what the optimizer multiplies the gradient by
    (FusedAdam grad_scales).  The pack is a copy per parameter into a preallocated bucket -- or nothing at all for a gradient its
    producer already wrote into the bucket (slot()) -- so nothing allocates after the first iteration (HIP-graph friendly).

    groups: list of lists of parameters (one list per chunk, in the order the next iteration needs them).  local_weight: views this
    rank renders; equal_shards: every rank has the same local_weight (then no weighting traffic at all: plain sum).

    sparse: None, or one bool per chunk.  A sparse chunk is exchanged tile by tile (TILE_FLOATS floats):
        compute_flags()   after pack(): which tiles of this rank's bucket are non-zero (one HIP launch; capturable in a HIP graph)
        start()           MAX all-reduce of the flag bytes (asynchronous), beside the dense chunks' SUM all-reduces
        send(k)           union list (one launch) -> its size read by the host -> gather -> SUM all-reduce of the compacted tiles (asynchronous);
                          runs on a SIDE stream that waits for the flags only, so whatever the caller enqueued on the main stream in between
                          (the next iteration's geometry stage) is not waited for
        wait(k)           scatter the sums back into the dense bucket; .grad views as for a dense chunk
    When the union exceeds `sparse_max_fraction` of the tiles the dense bucket is all-reduced instead (every rank sees the same count,
    so every rank takes the same branch).  Uneven shards and chunks whose parameters are not whole tiles fall back to dense.
    An entry of `sparse` may also be the string 'auto': the tile machinery costs ~0.1 ms per iteration of its own (flags, union list, host
    read, gather, scatter), so it only pays when few tiles are touched -- 'auto' PROBES the union every `probe_every` rounds (the first
    included) and runs the rounds in between sparse if the probe found at most `auto_fraction` of the tiles touched, plainly dense otherwise
    (no flags, no extra collective).  Measured on the benchmark views: one 512^2 view touches 24 % of the 64-texel tiles of a 1024^2 texture,
    the union of the batch's eight views 49 % (18 % of the TEXELS; tools/tile_fraction_probe.py) -- and with the reference's mip-mapped
    lookups every tile would be touched -- so 'auto' settles on dense there."""

    def __init__(self, groups, world_size=None, group=None, local_weight=1, equal_shards=True, sparse=None, tile_floats=TILE_FLOATS,
                 sparse_max_fraction=0.5, auto_fraction=0.25, probe_every=64):
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
        self.tile_floats = int(tile_floats)
        self.sparse_max_fraction = float(sparse_max_fraction)
        sparse = list(sparse) if sparse is not None else [False] * len(self.groups)
        sparse = [False if s in (False, None, 'dense') else s for s in sparse]
        self.auto = [s == 'auto' for s in sparse]
        self.auto_fraction, self.probe_every, self._round, self._flags_round = float(auto_fraction), int(probe_every), 0, -1
        self.sparse = [bool(s) and self.equal_shards and all(p.numel() % self.tile_floats == 0 for p in g) for s, g in zip(sparse, self.groups)]
        self._sp = {}
        for k, (s, b) in enumerate(zip(self.sparse, self.buckets)):
            if s:
                n_tiles = b.numel() // self.tile_floats
                dev = b.device
                self._sp[k] = {'n_tiles': n_tiles,
                               'flags': torch.zeros(n_tiles, dtype=torch.uint8, device=dev),
                               'list': torch.zeros(max(n_tiles, 1), dtype=torch.int32, device=dev),
                               'count': torch.zeros(1, dtype=torch.int32, device=dev),
                               'count_host': (torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else torch.zeros(1, dtype=torch.int32)),
                               'compact': torch.zeros_like(b),          # worst case: every tile (288 GB of HBM: 38 MB is nothing)
                               'flags_handle': None, 'state': 'idle', 'mode': 'dense', 'tiles': 0,
                               'use': True, 'auto_sparse': False}       # this round goes through the tiles / what the last probe of 'auto' decided
        self.extra_flags = {}               # chunk -> uint8 flags OR-ed into this rank's (trainer union_views: a one-rank run that moves the bytes of a several-rank one)
        self._side, self._ev_start = None, None      # the side stream of send() and the point of the main stream it is ordered behind (GPU tensors)
        self._sends = self.active and self.world > 1
        self.bytes_dense = sum(b.numel() for b in self.buckets) * 4
        self.bytes_per_step = self.bytes_dense if self._sends else 0     # what the last start()/send() round put into collectives
        self._bytes_round = 0

    # ------------------------------------------------------------------------------------------------------------------ layout
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
                    if dst.is_cuda:     # an elementwise KERNEL, not copy_(): a device-to-device memcpy node stalls a replayed HIP graph for ~30 us on this runtime
                        torch.mul(p.grad.reshape(-1), 1.0, out=dst)
                    else:
                        dst.copy_(p.grad.reshape(-1))
            if not self.equal_shards:
                b[:n].mul_(self.local_weight)
                b[n:].fill_(self.local_weight)

    def compute_flags(self):
        """Sparse chunks: flag the non-zero tiles of this rank's bucket (after pack(); one launch per sparse chunk, capturable)."""
        self._flags_round = self._round          # (start() checks that the flags belong to the round it sends)
        for k, sp in self._sp.items():
            sp['use'] = (not self.auto[k]) or sp['auto_sparse'] or self._round % self.probe_every == 0
            if not sp['use']:
                continue
            _TileOps.flags(self.buckets[k], sp['n_tiles'], self.tile_floats, sp['flags'])
            if self.extra_flags.get(k) is not None:         # (one-rank dry runs: the tiles the other ranks would have touched)
                torch.maximum(sp['flags'], self.extra_flags[k], out=sp['flags'])

    # ------------------------------------------------------------------------------------------------------------- collectives
    def _all_reduce(self, t, op):
        return dist.all_reduce(t, op=op, group=self.group, async_op=True)

    def start(self, skip_single=True):
        """Launch what can be launched right after the backward pass, in chunk order: the SUM all-reduce of every dense chunk, the MAX
        all-reduce of every sparse chunk's tile flags (all asynchronous).  With one rank (and skip_single) nothing is sent."""
        self.handles = [None] * len(self.buckets)
        self._bytes_round = 0
        live = self.active and not (self.world == 1 and skip_single)
        if self._sp and self._flags_round != self._round:
            # pack() -> start() without compute_flags(): stale flags would leave touched tiles un-reduced and the ranks would
            # diverge silently -- flag the tiles of THIS round here (one launch per sparse chunk)
            self.compute_flags()
        self._round += 1
        for k, b in enumerate(self.buckets):
            if k in self._sp and self._sp[k]['use']:
                sp = self._sp[k]
                sp['state'], sp['flags_handle'] = 'flagged', None
                if live:
                    sp['flags_handle'] = self._all_reduce(sp['flags'], dist.ReduceOp.MAX)
                    self._bytes_round += sp['flags'].numel()
            else:
                if k in self._sp:
                    self._sp[k]['state'], self._sp[k]['mode'] = 'idle', 'dense'
                if live:
                    self.handles[k] = self._all_reduce(b, dist.ReduceOp.SUM)
                    self._bytes_round += b.numel() * 4
        self._live = live
        if self._sp and self.buckets[0].is_cuda:
            # send() runs on a side stream ordered behind THIS point of the main stream (the buckets and flags are complete here), not
            # behind whatever the caller enqueues between start() and send()
            self._ev_start = torch.cuda.Event()
            self._ev_start.record()
        self.bytes_per_step = self._bytes_round        # (send() adds what a sparse chunk's second stage puts on the wire)

    def send(self, k):
        """Second stage of a sparse chunk (a no-op for a dense one): the union's tile list, its size to the host, the gather and the
        asynchronous SUM all-reduce of the compacted tiles.  Blocks the host until the flags of chunk k have been reduced -- not until
        the main stream has drained: GPU work enqueued after start() keeps running."""
        sp = self._sp.get(k)
        if sp is None or sp['state'] != 'flagged':
            return
        b = self.buckets[k]
        cuda = b.is_cuda
        if cuda and self._side is None:
            # (normal priority, as in round 5: a HIGH-priority side stream here -- tried in round 6 to keep it off the main stream's hardware queue --
            # went with a sporadic stall of the tile-sparse schedule, one run in three; the dense exchange has no side stream at all)
            self._side = torch.cuda.Stream(device=b.device)
        if cuda:
            self._side.wait_event(self._ev_start)        # the main stream as of start(), not as of now
        ctx = torch.cuda.stream(self._side) if cuda else _null_context()
        with ctx:
            if sp['flags_handle'] is not None:
                sp['flags_handle'].wait()
            _TileOps.plan(sp['flags'], sp['n_tiles'], sp['list'], sp['count'])
            if cuda:
                sp['count_host'].copy_(sp['count'], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
                ev.synchronize()
                n = int(sp['count_host'][0])
            else:
                n = int(sp['count'].item())
            sp['tiles'] = n
            limit = self.auto_fraction if self.auto[k] else self.sparse_max_fraction
            sp['auto_sparse'] = n <= limit * sp['n_tiles']
            if n > limit * sp['n_tiles']:
                sp['mode'] = 'dense'
                if self._live:
                    self.handles[k] = self._all_reduce(b, dist.ReduceOp.SUM)
                    self._bytes_round += b.numel() * 4
            else:
                sp['mode'] = 'sparse'
                _TileOps.move(b, sp['compact'], sp['list'], sp['count'], sp['n_tiles'], self.tile_floats, gather=True)
                if self._live and n > 0:
                    self.handles[k] = self._all_reduce(sp['compact'][:n * self.tile_floats], dist.ReduceOp.SUM)
                    self._bytes_round += n * self.tile_floats * 4
        sp['state'] = 'sent'
        self.bytes_per_step = self._bytes_round

    def wait(self, k):
        """Chunk k has arrived: point the .grad of its parameters at the reduced bucket (views, no copy).  Returns the factor the
        optimizer must apply to these gradients (float, or a 0-dim device tensor for uneven shards)."""
        sp = self._sp.get(k)
        if sp is not None and sp['state'] == 'flagged':
            self.send(k)
        h = self.handles[k] if self.handles else None
        if h is not None:
            h.wait()
        b = self.buckets[k]
        if sp is not None and sp['state'] == 'sent':
            if b.is_cuda:
                torch.cuda.current_stream(b.device).wait_stream(self._side)    # (the gather / a one-rank run: no collective handle to wait on)
            if sp['mode'] == 'sparse':
                _TileOps.move(b, sp['compact'], sp['list'], sp['count'], sp['n_tiles'], self.tile_floats, gather=False)
            sp['state'] = 'idle'
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

    def report(self):
        """What the last round sent: {mode, bytes_dense, bytes_sent, tiles_touched, tiles_total} (bench.py config.exchange)."""
        modes = [self._sp[k]['mode'] if k in self._sp else 'dense' for k in self.chunks()]
        return {'mode': 'sparse' if 'sparse' in modes else 'dense', 'chunk_modes': modes,
                'policy': ['auto' if a else ('sparse' if s else 'dense') for a, s in zip(self.auto, self.sparse)], 'chunk_bytes_dense': [b.numel() * 4 for b in self.buckets],
                'bytes_dense': self.bytes_dense, 'bytes_sent': int(self.bytes_per_step),
                'tiles_touched': int(sum(sp['tiles'] for sp in self._sp.values())), 'tiles_total': int(sum(sp['n_tiles'] for sp in self._sp.values())),
                'tile_bytes': self.tile_floats * 4}


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
