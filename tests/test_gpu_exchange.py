"""The device side of the tile-sparse gradient exchange (csrc/exchange.hip) against plain torch indexing: flags, the ascending list
of the flagged tiles, gather and scatter -- and a whole sparse round on GPU tensors (one rank, no collective) leaves the bucket as it was."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n_tiles,tile_floats', [(49152, 192), (1000, 192), (7, 64), (3000, 4)])
def test_tile_kernels_match_torch_indexing(n_tiles, tile_floats, dev):
    from nvdiffrecmc_amd.parallel import _TileOps
    g = torch.Generator().manual_seed(n_tiles)
    dense = torch.zeros(n_tiles * tile_floats)
    touched = torch.rand(n_tiles, generator=g) < 0.2
    vals = torch.randn(n_tiles, tile_floats, generator=g) * (torch.rand(n_tiles, tile_floats, generator=g) < 0.05)
    dense.view(n_tiles, tile_floats)[touched] = vals[touched]
    dense.view(n_tiles, tile_floats)[5 % n_tiles, 1] = float('nan')        # a NaN must travel
    dense.view(n_tiles, tile_floats)[3 % n_tiles] = 0.0
    dense.view(n_tiles, tile_floats)[3 % n_tiles, 0] = -0.0               # minus zero is zero
    d = dense.to(dev)
    flags = torch.empty(n_tiles, dtype=torch.uint8, device=dev)
    _TileOps.flags(d, n_tiles, tile_floats, flags)
    want = (dense.view(n_tiles, tile_floats) != 0).any(1)
    assert torch.equal(flags.cpu().bool(), want)
    lst = torch.full((n_tiles,), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _TileOps.plan(flags, n_tiles, lst, cnt)
    k = int(cnt.item())
    idx = want.nonzero().view(-1)
    assert k == idx.numel() and torch.equal(lst[:k].cpu().long(), idx)
    compact = torch.full((n_tiles * tile_floats,), 7.0, device=dev)
    _TileOps.move(d, compact, lst, cnt, n_tiles, tile_floats, gather=True)
    got = compact[:k * tile_floats].view(k, tile_floats).cpu()
    ref = dense.view(n_tiles, tile_floats)[idx]
    assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(ref, nan=123.0))
    assert (compact[k * tile_floats:] == 7.0).all()                         # nothing written behind the count
    back = torch.zeros_like(d)
    _TileOps.move(back, compact * 2.0, lst, cnt, n_tiles, tile_floats, gather=False)
    assert torch.equal(torch.nan_to_num(back.cpu(), nan=123.0), torch.nan_to_num(dense * 2.0, nan=123.0))


def test_sparse_round_without_a_collective_is_the_identity(dev):
    from nvdiffrecmc_amd.parallel import GradientExchange
    a = torch.nn.Parameter(torch.zeros(256, 256, 3, device=dev))
    b = torch.nn.Parameter(torch.zeros(16, 16, 3, device=dev))
    ex = GradientExchange([[b], [a]], 1, sparse=[False, True])
    assert ex.sparse == [False, True]
    a.grad = ex.slot(a)
    a.grad[10:20, 30:90] = torch.randn(10, 60, 3, device=dev)
    b.grad = torch.randn(16, 16, 3, device=dev)
    want_a, want_b = a.grad.clone(), b.grad.clone()
    ex.pack(); ex.compute_flags(); ex.start()
    assert ex.wait(0) == 1.0 and torch.equal(b.grad, want_b)
    ex.send(1)
    assert ex.wait(1) == 1.0 and torch.equal(a.grad, want_a)
    rep = ex.report()
    assert rep['mode'] == 'sparse' and rep['tiles_touched'] == int((want_a.view(-1, 192) != 0).any(1).sum()) and rep['bytes_sent'] == 0
    # an empty gradient: zero tiles, nothing gathered, nothing scattered
    a.grad.zero_()
    ex.pack(); ex.compute_flags(); ex.start(); ex.wait(0); ex.send(1); ex.wait(1)
    assert ex.report()['tiles_touched'] == 0 and not a.grad.any()
