"""LBVH build + traversal on the GPU vs the oracle's brute force: bit-exact visibility and closest hits."""
import math

import pytest
import torch

from oracle import oracle as orc
from nvdiffrecmc_amd import scene as sc
from tests.gpu_util import make_ctx

pytestmark = pytest.mark.gpu
NT = orc.max_threads()


def _rays(n, seed, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    ro = torch.randn(n, 3, generator=g) * scale
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    return ro.contiguous(), rd.contiguous()


@pytest.mark.parametrize('mesh_name', ['bob', 'spot'])
def test_visibility_bit_exact_vs_bruteforce(mesh_name, dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    ctx = make_ctx(mesh, dev)
    info = ctx.bvh_info()
    assert info['n_tris'] == mesh['t_pos_idx'].shape[0] and 1 <= info['height'] <= 64
    ro, rd = _rays(200000, 1)
    # add axis-parallel and surface-grazing rays (edge cases of the slab test)
    rd[:3000] = torch.eye(3).repeat(1000, 1)
    v = mesh['v_pos']
    ro[3000:6000] = v[torch.randint(0, v.shape[0], (3000,), generator=torch.Generator().manual_seed(2))]
    ref = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    got, cnt = ou.trace_visibility(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got.cpu(), ref), '%d of %d rays differ' % (int((got.cpu() != ref).sum()), ro.shape[0])
    assert 0.05 < ref.float().mean().item() < 0.95
    assert cnt[0].item() > 0 and cnt[1].item() > 0


def test_closest_hit_vs_bruteforce(dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    ctx = make_ctx(mesh, dev)
    mv, _, _ = sc.camera(2, 8)
    ro, rd = sc.primary_rays(mv, 96, 96)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    t, tri, uv = orc.closest(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    gt, gtri, guv = ou.trace_closest(ctx, ro.to(dev), rd.to(dev))
    gt, gtri, guv = gt.cpu(), gtri.cpu(), guv.cpu()
    assert torch.equal((gtri >= 0), (tri >= 0))
    hit = tri >= 0
    assert hit.float().mean().item() > 0.1
    assert torch.equal(gt[hit], t[hit])                       # same predicate, same arithmetic: identical distances
    same = gtri == tri                                        # ties between coplanar neighbours may pick either triangle
    assert same[hit].float().mean().item() > 0.999
    assert torch.equal(guv[hit & same], uv[hit & same])


def test_refit_equals_rebuild_and_single_triangle(dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('spot')
    v, t = mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v, t, rebuild=1)
    g = torch.Generator().manual_seed(5)
    v2 = (mesh['v_pos'] * 1.1 + 0.01 * torch.randn(mesh['v_pos'].shape, generator=g)).contiguous()
    ou.optix_build_bvh(ctx, v2.to(dev), t, rebuild=0)          # refit the old topology to moved vertices
    ro, rd = _rays(50000, 3, 0.35)
    ref = orc.visibility(v2, mesh['t_pos_idx'], ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    ou.optix_build_bvh(ctx, v2.to(dev), t, rebuild=1)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    with pytest.raises(RuntimeError, match='refit'):
        ou.optix_build_bvh(ctx, v2.to(dev), t[:100].contiguous(), rebuild=0)
    # degenerate sizes: one triangle, two triangles
    for nt in (1, 2):
        tv = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
        tt = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)[:nt].contiguous()
        ou.optix_build_bvh(ctx, tv.to(dev), tt.to(dev), rebuild=1)
        ro2 = torch.tensor([[0.2, 0.2, 1.0], [0.2, 0.2, 1.0], [-1.0, 0.2, 0.2]])
        rd2 = torch.tensor([[0.0, 0, -1.0], [0.0, 0, 1.0], [1.0, 0, 0]])
        ref2 = orc.visibility(tv, tt, ro2, rd2)
        assert torch.equal(ou.trace_visibility(ctx, ro2.to(dev), rd2.to(dev)).cpu(), ref2)


def test_build_rejects_empty_mesh_like_the_reference(dev):
    from nvdiffrecmc_amd import optixutils as ou
    ctx = ou.OptiXContext()
    with pytest.raises(AssertionError, match='empty training triangle mesh'):
        ou.optix_build_bvh(ctx, torch.zeros(3, 3, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev), 1)


def test_large_mesh_build_and_trace(dev):
    """DMTet-sized stand-in: bob subdivided twice (171k triangles)."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    v, t = sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], 2)
    assert t.shape[0] == 10688 * 16
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    ro, rd = _rays(20000, 9)
    ref = orc.visibility(v, t, ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
