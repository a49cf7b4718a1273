"""Bilateral denoiser (LDS-tiled HIP) vs the oracle, the reference vectors and autograd of the authors' torch filter."""
import pytest
import torch

from oracle import oracle as orc, renderutils_ref as rr
from tests.util import load_npz, assert_close
from tools import make_golden as mg

pytestmark = pytest.mark.gpu
NT = orc.max_threads()
# fast exp / reciprocal and x^128 by squaring in the kernel vs expf / division / powf in the reference: 2e-5
RTOL = 2e-5


@pytest.mark.parametrize('case', mg.DN_CASES, ids=[c[0] for c in mg.DN_CASES])
def test_denoiser_vs_reference_vectors(case, dev):
    from nvdiffrecmc_amd.optixutils.ops import _bilateral_denoiser_func
    name, N, H, W, sigma, seed = case
    gold = load_npz('denoiser_reference.npz')[name]
    x, col, nrm, zdz, og = mg.denoiser_inputs(N, H, W, seed)
    xd = x.to(dev)
    col_d = xd[..., 0:3].clone().requires_grad_(True)
    out = _bilateral_denoiser_func.apply(col_d, nrm.to(dev), xd[..., 6:8], sigma)   # zdz as a strided view
    assert_close(out.detach(), gold['out'], RTOL, floor=1e-4, what='out')
    out.backward(og.to(dev))
    assert_close(col_d.grad, gold['col_grad'], RTOL, floor=1e-4, what='col_grad')


@pytest.mark.parametrize('N,H,W,sigma', [(1, 67, 45, 2.0), (2, 33, 70, 1.0), (1, 8, 8, 2.0), (1, 40, 40, 6.0)])
def test_denoiser_vs_oracle_ragged(N, H, W, sigma, dev):
    """Ragged extents (tile edges), image smaller than the filter, and a radius too large for the LDS tile."""
    from nvdiffrecmc_amd import optixutils as ou
    x, col, nrm, zdz, og = mg.denoiser_inputs(N, H, W, 31)
    zdz = zdz.contiguous()
    ref = orc.bilateral_fwd(col, nrm, zdz, sigma, n_threads=NT)
    refg = orc.bilateral_bwd(col, nrm, zdz, sigma, og, n_threads=NT)
    cd = col.to(dev).requires_grad_(True)
    out = ou.ops._bilateral_denoiser_func.apply(cd, nrm.to(dev), zdz.to(dev), sigma)
    assert_close(out.detach(), ref, RTOL, floor=1e-4)
    out.backward(og.to(dev))
    assert_close(cd.grad, refg, RTOL, floor=1e-4)


def test_denoiser_module_matches_authors_torch_filter(dev):
    """BilateralDenoiser.forward on the 8-channel cat tensor (denoiser.py:27-31) incl. the division, fwd and bwd."""
    from nvdiffrecmc_amd.denoiser import BilateralDenoiser
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 36, 30, 8, generator=g)
    x[..., 3:6] = x[..., 3:6] * 2 - 1
    tgt = torch.rand(1, 36, 30, 3, generator=g)
    xr = x.clone().requires_grad_(True)
    nrm = xr[..., 3:6] / torch.sqrt(torch.clamp((xr[..., 3:6] ** 2).sum(-1, keepdim=True), min=1e-20))
    ref = rr.bilateral_denoiser_torch(xr[..., 0:3], nrm.detach(), xr[..., 6:8].detach(), 2.0)
    torch.nn.functional.mse_loss(ref, tgt).backward()
    xd = x.to(dev).requires_grad_(True)
    den = BilateralDenoiser(influence=1.0)
    assert den.sigma == 2.0 and den.N == 11
    out = den.forward(xd)
    torch.nn.functional.mse_loss(out, tgt.to(dev)).backward()
    assert_close(out.detach(), ref.detach(), RTOL, floor=1e-4)
    assert_close(xd.grad[..., 0:3], xr.grad[..., 0:3], 1e-3, floor=1e-2 * xr.grad.abs().max().item())
    den.set_influence(0.0)
    assert den.sigma == 0.0001 and den.N == 3


def test_denoiser_full_size_properties(dev):
    """512x512 (the benchmark size): constant colour is reproduced exactly where weight exists; the
    backward kernel is the exact transpose of the forward one (<fwd(c), g> == <c, bwd(g)>)."""
    from nvdiffrecmc_amd.optixutils.ops import _bilateral_denoiser_func
    g = torch.Generator().manual_seed(1)
    N, H, W = 1, 512, 512
    nrm = torch.nn.functional.normalize(torch.rand(N, H, W, 3, generator=g) + 0.5, dim=-1).to(dev)
    zdz = torch.rand(N, H, W, 2, generator=g).to(dev)
    const = torch.full((N, H, W, 3), 0.37, device=dev)
    o = _bilateral_denoiser_func.apply(const, nrm, zdz, 2.0)
    ok = o[..., 3] > 1e-3
    assert_close((o[..., :3] / o[..., 3:4])[ok], const[ok], 1e-5)
    c = torch.rand(N, H, W, 3, generator=g).to(dev).requires_grad_(True)
    gr = torch.rand(N, H, W, 4, generator=g).to(dev)
    out = _bilateral_denoiser_func.apply(c, nrm, zdz, 2.0)
    lhs = (out[..., :3] * gr[..., :3]).sum().double()
    out.backward(gr)
    rhs = (c.detach() * c.grad).sum().double()
    assert abs(lhs.item() - rhs.item()) < 1e-4 * abs(lhs.item())


@pytest.mark.parametrize('sigma', [2.0, 6.0])
def test_denoiser_background_early_out(sigma, dev):
    """Pixels with an exactly-zero normal (everything the rasteriser did not cover) have zero weight on every tap; the
    kernel skips whole workgroups / wavefronts of them.  Mixed, fully empty and fully covered tiles vs the oracle."""
    from nvdiffrecmc_amd import optixutils as ou
    N, H, W = 1, 96, 160
    x, col, nrm, zdz, og = mg.denoiser_inputs(N, H, W, 17)
    nrm = nrm.clone()
    nrm[:, :, 70:] = 0.0          # columns 70.. : whole 32x8 workgroups empty from x = 96 on, mixed ones before
    nrm[:, 40:50, :] = 0.0        # an empty band: wavefront-level skips inside covered workgroups
    nrm[:, 3, 5] = 0.0            # a single hole
    zdz = zdz.contiguous()
    ref = orc.bilateral_fwd(col, nrm, zdz, sigma, n_threads=NT)
    refg = orc.bilateral_bwd(col, nrm, zdz, sigma, og, n_threads=NT)
    cd = col.to(dev).requires_grad_(True)
    out = ou.ops._bilateral_denoiser_func.apply(cd, nrm.to(dev), zdz.to(dev), sigma)
    assert_close(out.detach(), ref, RTOL, floor=1e-4)
    out.backward(og.to(dev))
    assert_close(cd.grad, refg, RTOL, floor=1e-4)
    empty = (nrm == 0).all(-1)
    assert torch.equal(out.detach().cpu()[empty][:, :3], torch.zeros(int(empty.sum()), 3))
    assert torch.equal(out.detach().cpu()[empty][:, 3], torch.full((int(empty.sum()),), 1e-4))
    assert torch.equal(cd.grad.cpu()[empty], torch.zeros(int(empty.sum()), 3))


@pytest.mark.parametrize('N,H,W,sigma', [(1, 70, 45, 1.0), (2, 96, 130, 2.0), (1, 40, 40, 4.0)])
def test_pair_filter_equals_two_single_calls(N, H, W, sigma, dev):
    """Two images with the same guides filtered in one pass (weights evaluated once per tap; the diffuse and the specular light of
    shade(), render.py:120-121) are bit-identical to two single calls, forward and backward -- ragged extents, a background region
    (zero normals), the tiled and (sigma = 4: the three-plane tile no longer fits) the untiled path."""
    from nvdiffrecmc_amd import optixutils as ou
    g = torch.Generator().manual_seed(7)
    ca, cb = torch.rand(N, H, W, 3, generator=g), torch.rand(N, H, W, 3, generator=g) * 3.0
    nrm = torch.nn.functional.normalize(torch.randn(N, H, W, 3, generator=g), dim=-1)
    nrm[:, : H // 3, : W // 2] = 0.0                                   # background
    zdz = torch.rand(N, H, W, 2, generator=g) + 0.1
    ga, gb = torch.rand(N, H, W, 4, generator=g), torch.rand(N, H, W, 4, generator=g)
    ca, cb, nrm, zdz, ga, gb = (t.to(dev) for t in (ca, cb, nrm, zdz, ga, gb))
    xa, xb = ca.clone().requires_grad_(True), cb.clone().requires_grad_(True)
    oa = ou.ops._bilateral_denoiser_func.apply(xa, nrm, zdz, sigma)
    ob = ou.ops._bilateral_denoiser_func.apply(xb, nrm, zdz, sigma)
    torch.autograd.backward([oa, ob], [ga, gb])
    ya, yb = ca.clone().requires_grad_(True), cb.clone().requires_grad_(True)
    pa, pb = ou.ops._bilateral_denoiser_pair_func.apply(ya, yb, nrm, zdz, sigma)
    torch.autograd.backward([pa, pb], [ga, gb])
    assert torch.equal(pa, oa) and torch.equal(pb, ob)
    assert torch.equal(ya.grad, xa.grad) and torch.equal(yb.grad, xb.grad)
