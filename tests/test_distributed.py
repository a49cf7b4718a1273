"""The N > 1 path on the CPU: world_size-2 gloo processes, one view per rank, ONE flat all-reduce of the shared-parameter
gradients (nvdiffrecmc_amd/parallel.py).  The renderer inside is the CPU oracle (the HIP kernels need a GPU); what is
checked is the sharding + reduction logic the GPU job uses unchanged."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nvdiffrecmc_amd.parallel import shard_views, allreduce_gradients


def test_shard_views():
    assert shard_views(8, 3, 8) == [3]
    assert shard_views(8, 1, 2) == [4, 5, 6, 7]
    assert shard_views(5, 1, 2) == [3, 4] and shard_views(5, 0, 2) == [0, 1, 2]
    assert sorted(sum((shard_views(7, r, 4) for r in range(4)), [])) == list(range(7))
    assert allreduce_gradients([torch.nn.Parameter(torch.ones(3))]) == 0      # no process group: no-op


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _light_grad_of_view(view, offset, res=20, n=2):
    """d(sum diff + sum spec) / d light for one view through the oracle (stands in for the HIP backward)."""
    from oracle import oracle as orc, scene_cpu
    inp = scene_cpu.make_inputs('bob', res, res, n, view=view, n_views=2, probe_res=16, n_threads=2)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    ones = torch.ones(1, res, res, 3)
    b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, n_samples_x=n, rnd_seed=3, diff_grad=ones, spec_grad=ones,
                      pixel_index_offset=offset, n_threads=1)
    return b['light_grad'], b['gb_ks_grad'].sum(dim=(0, 1, 2))


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    res = 20
    view = shard_views(world, rank, world)[0]
    lg, ksg = _light_grad_of_view(view, rank * res * res)
    light = torch.nn.Parameter(torch.zeros(16, 16, 3))
    ks = torch.nn.Parameter(torch.zeros(3))
    unused = torch.nn.Parameter(torch.zeros(5))             # a parameter without gradient must not break the bucket layout
    light.grad, ks.grad = lg.clone(), ksg.clone()
    nbytes = allreduce_gradients([light, ks, unused], world)
    # numpy copies travel by value (a torch tensor is passed as a file descriptor that dies with this process)
    q.put((rank, light.grad.numpy().copy(), ks.grad.numpy().copy(), unused.grad.numpy().copy(), nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_batch_mean():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    got = [(r, torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c), n) for r, a, b, c, n in got]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: mean over the two views, each seeded as slice r of the batch launch
    l0, k0 = _light_grad_of_view(0, 0)
    l1, k1 = _light_grad_of_view(1, 20 * 20)
    for rank, lg, ksg, ug, nbytes in got:
        assert torch.allclose(lg, (l0 + l1) / 2, rtol=1e-5, atol=1e-7)
        assert torch.allclose(ksg, (k0 + k1) / 2, rtol=1e-5, atol=1e-6)
        assert torch.equal(ug, torch.zeros(5))
        assert nbytes == (16 * 16 * 3 + 3 + 5) * 4
    assert not torch.equal(got[0][1], torch.zeros_like(got[0][1]))


def _worker_uneven(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    views = shard_views(3, rank, world)                  # rank 0: views 0, 1 -- rank 1: view 2
    # each rank's gradient is the gradient of ITS mean over its views (what loss.backward() of a local batch yields)
    g = torch.stack([torch.full((4,), float(10 ** v)) for v in views]).mean(0)
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = g.clone()
    allreduce_gradients([p], world, local_weight=len(views))
    q.put((rank, p.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_keep_the_batch_mean():
    """3 views on 2 ranks: the plain average of the two local means would weight view 2 twice; the weighted bucket
    reproduces the mean over the three views."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.full((4,), (1.0 + 10.0 + 100.0) / 3.0)
    for rank, g in got:
        assert torch.allclose(torch.from_numpy(g), want, rtol=1e-6)


# ---------------------------------------------------------------------------------------------- the chunked exchange (round 4)
def _worker_exchange(rank, world, port, q, n_views):
    """The reference-sized parameter set (three 1024^2 textures = 37.7 MB, the probe, vertices) through GradientExchange: two chunks,
    the 'update' of chunk 0 done before chunk 1 is waited for, gradients read from the bucket views."""
    from nvdiffrecmc_amd.parallel import GradientExchange
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    views = shard_views(n_views, rank, world)
    shapes = [(1024, 1024, 3), (1024, 1024, 3), (1024, 1024, 3), (256, 256, 3), (5344, 3)]
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    # the gradient of THIS rank's mean over its views; view v contributes (v + 1) * pattern_k
    local = sum(float(v + 1) for v in views) / len(views)
    even = n_views % world == 0
    ex = GradientExchange([params[0:2], params[2:]], world, local_weight=len(views), equal_shards=even)
    for k, p in enumerate(params[:-1]):
        if k in (0, 2):                                     # produced inside the bucket (the trainer's texture gradients): nothing to pack
            p.grad = ex.slot(p)
            p.grad.fill_(local * (k + 1))
        else:
            p.grad = torch.full(p.shape, local * (k + 1))
    params[-1].grad = None                                  # a parameter without gradient: zeros in the bucket
    before = [b.clone() for b in ex.buckets]
    ex.pack()
    n0 = params[0].numel()
    assert torch.equal(ex.buckets[0][:n0], before[0][:n0]) or not even        # a resident gradient is left where it is (weighted: scaled in place)
    ex.start()
    out = []
    for k in ex.chunks():
        f = ex.wait(k)
        for p in ex.groups[k]:
            assert p.grad.data_ptr() >= ex.buckets[k].data_ptr()           # a view into the bucket, not a copy
            out.append(float((p.grad * f).double().mean()))
    q.put((rank, out, ex.bytes_per_step))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_views', [2, 3])
def test_chunked_exchange_of_the_reference_sized_bucket(n_views):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q, n_views)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    mean_view = sum(float(v + 1) for v in range(n_views)) / n_views       # the batch mean over ALL views, however they were dealt
    for rank, out, nbytes in got:
        want = [mean_view * (k + 1) for k in range(4)] + [0.0]
        assert len(out) == 5
        for a, b in zip(out, want):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (rank, out, want)
        n = 3 * 1024 * 1024 * 3 + 256 * 256 * 3 + 5344 * 3
        assert nbytes == (n + (0 if n_views % world == 0 else 2)) * 4        # >= 38 MB; uneven shards carry one weight per chunk
        assert nbytes >= 38_000_000


# ---------------------------------------------------------------------------------------------- the tile-sparse exchange (round 5)
def _worker_sparse(rank, world, port, q, density):
    """kd / ks / normal textures with gradient at a few texels per rank (a nearest-texel lookup's adjoint) through the exchange in both
    modes: probe + vertices first (dense), then the textures dense resp. tile-sparse."""
    from nvdiffrecmc_amd.parallel import GradientExchange, TILE_FLOATS
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    R = 128
    shapes = [(R, R, 3), (R, R, 3), (R, R, 3), (16, 16, 3), (37, 3)]
    gen = torch.Generator().manual_seed(100 + rank)
    grads = []
    for s in shapes[:3]:
        g = torch.zeros(*s)
        n = int(density * R * R)
        for _ in range(4):                                  # four runs of neighbouring texels: what a view's pixels touch is clustered
            first = int(torch.randint(0, R * R - n // 4, (1,), generator=gen))
            g.view(-1, 3)[first:first + n // 4] = torch.randn(n // 4, 3, generator=gen)
        grads.append(g)
    grads += [torch.randn(16, 16, 3, generator=gen), torch.randn(37, 3, generator=gen)]
    results = {}
    for mode in ('dense', 'sparse'):
        params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
        ex = GradientExchange([params[3:], params[:3]], world, local_weight=1, equal_shards=True, sparse=[False, mode == 'sparse'])
        assert ex.sparse == [False, mode == 'sparse']
        for p, g in zip(params[:3], grads[:3]):             # born in the bucket, as the trainer's lookup adjoint leaves them
            p.grad = ex.slot(p)
            p.grad.copy_(g)
        params[3].grad, params[4].grad = grads[3].clone(), grads[4].clone()
        ex.pack()
        ex.compute_flags()
        ex.start()
        f = ex.wait(0)                                      # probe + vertices: the next iteration's geometry stage could start here
        early = [(p.grad * f).clone() for p in params[3:]]
        ex.send(1)
        f = ex.wait(1)
        tex = [(p.grad * f).clone() for p in params[:3]]
        results[mode] = (early + tex, ex.report())
    dd, ds = results['dense'][0], results['sparse'][0]
    same = all(torch.equal(a, b) for a, b in zip(dd, ds))   # the same addends through the same collective: bit for bit
    rep = results['sparse'][1]
    q.put((rank, same, rep, [t.double().sum().item() for t in ds], results['dense'][1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('density', [0.04, 0.9], ids=['sparse_4pct', 'falls_back_to_dense'])
def test_tile_sparse_exchange_equals_the_dense_one(density):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, world, port, q, density)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, rep, sums, rep_dense in got:
        assert same, 'sparse and dense exchange differ on rank %d' % rank
        assert rep['bytes_dense'] == rep_dense['bytes_dense'] == rep_dense['bytes_sent'] == (3 * 128 * 128 * 3 + 16 * 16 * 3 + 37 * 3) * 4
        assert rep['tiles_total'] == 3 * 128 * 128 // 64
        if density < 0.5:
            assert rep['mode'] == 'sparse' and 0 < rep['tiles_touched'] < 0.5 * rep['tiles_total']
            # flags (1 B per tile) + the dense probe / vertex chunk + the touched tiles
            assert rep['bytes_sent'] == rep['tiles_total'] + (16 * 16 * 3 + 37 * 3) * 4 + rep['tiles_touched'] * 768
            assert rep['bytes_sent'] < 0.6 * rep['bytes_dense']
        else:
            assert rep['mode'] == 'dense' and rep['tiles_touched'] > 0.5 * rep['tiles_total']       # the union is large: the dense bucket travels
            assert rep['bytes_sent'] == rep['bytes_dense'] + rep['tiles_total']
    assert got[0][3] == got[1][3]                           # both ranks hold the same sums
    assert got[0][2]['tiles_touched'] == got[1][2]['tiles_touched']


def _worker_auto(rank, world, port, q, density):
    """exchange policy 'auto': the first round probes the union of the touched tiles; the rounds behind it run sparse (few tiles) or plainly dense."""
    from nvdiffrecmc_amd.parallel import GradientExchange
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    R = 128
    gen = torch.Generator().manual_seed(7 + rank)
    g = torch.zeros(R, R, 3)
    n = int(density * R * R)
    first = int(torch.randint(0, R * R - n, (1,), generator=gen))
    g.view(-1, 3)[first:first + n] = torch.randn(n, 3, generator=gen)
    tex, probe = torch.nn.Parameter(torch.zeros(R, R, 3)), torch.nn.Parameter(torch.zeros(8, 8, 3))
    ex = GradientExchange([[probe], [tex]], world, sparse=[False, 'auto'], probe_every=4)
    out = []
    for rnd in range(6):
        tex.grad = ex.slot(tex)
        tex.grad.copy_(g * (rnd + 1))
        probe.grad = torch.full((8, 8, 3), float(rank + 1))
        ex.pack(); ex.compute_flags(); ex.start()
        f0 = ex.wait(0)
        ex.send(1)
        f = ex.wait(1)
        rep = ex.report()
        out.append((rep['mode'], rep['bytes_sent'], float((tex.grad * f).double().sum()), float((probe.grad * f0).double().mean())))
        tex.grad.zero_()
    q.put((rank, out, ex.bytes_dense, float(g.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('density', [0.03, 0.6], ids=['few_tiles_sparse', 'many_tiles_dense'])
def test_auto_policy_probes_and_settles(density):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_auto, args=(r, world, port, q, density)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = got[0][3] + got[1][3]
    tiles = 128 * 128 // 64
    for rank, out, dense, _ in got:
        for rnd, (mode, sent, s, pm) in enumerate(out):
            assert abs(s - 0.5 * total * (rnd + 1)) <= 1e-5 * abs(total) * (rnd + 1)          # the batch mean whichever way the bytes went
            assert pm == 1.5
            probe_round = rnd % 4 == 0
            if density < 0.25:
                assert mode == 'sparse' and sent < 0.3 * dense                              # flags + touched tiles every round
            else:
                assert mode == 'dense' and sent == dense + (tiles if probe_round else 0)    # the probe costs its flag bytes, the other rounds nothing
