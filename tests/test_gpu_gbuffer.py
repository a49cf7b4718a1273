"""The G-buffer producer kernel (csrc/gbuffer.hip) against a torch restatement of render_layer's interpolation
(render/render.py:208-234, oracle/render_layer_ref.py) driven by the kernel's own rast / rast_db, and its primary visibility
against the oracle's brute-force closest hit."""
import pytest
import torch

from oracle import oracle as orc, render_layer_ref as rl
from nvdiffrecmc_amd import scene as sc
from tests.util import assert_close

pytestmark = pytest.mark.gpu
NT = orc.max_threads()
MESH_KEYS = ('v_pos', 't_pos_idx', 'v_nrm', 't_nrm_idx', 'v_tng', 't_tng_idx', 'v_tex', 't_tex_idx')


def _views(views, n_views, aspect=1.0):
    mvp, cam = [], []
    for k in views:
        mv, p, _ = sc.camera(k, n_views, aspect=aspect)
        mvp.append(p)
        cam.append(sc.camera_rays(mv, aspect=aspect))
    return torch.stack(mvp).contiguous(), torch.stack(cam).contiguous()


@pytest.mark.parametrize('mesh_name,H,W', [('bob', 96, 96), ('spot', 64, 112)])
def test_gbuffer_kernel_vs_render_layer_restatement(mesh_name, H, W, dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    md = {k: mesh[k].to(dev) for k in MESH_KEYS}
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, md['v_pos'], md['t_pos_idx'], 1)
    mvp, cam = _views([1, 6], 8, aspect=W / H)
    gb = {k: v.cpu() for k, v in ou.render_gbuffer(ctx, md, mvp.to(dev), cam.to(dev), (H, W)).items()}
    rast, rast_db = gb['rast'], gb['rast_db']
    assert rast.shape == (2, H, W, 4) and gb['gb_depth'].shape == (2, H, W, 2) and gb['gb_texc_deriv'].shape == (2, H, W, 4)
    covered = rast[..., 3] > 0
    assert 0.1 < covered.float().mean().item() < 0.6
    assert (rast[~covered] == 0).all() and (gb['gb_pos'][~covered] == 0).all()
    assert (gb['gb_depth'][~covered] == torch.tensor([1.0, 0.0])).all()         # render.py:230 on all-zero attributes: eps / eps

    # (1) primary visibility: the same triangle and barycentrics as the oracle's brute-force closest hit of the same rays
    for n in range(2):
        X = ((torch.arange(W) + 0.5) / W * 2 - 1)[None, :, None]
        Y = ((torch.arange(H) + 0.5) / H * 2 - 1)[:, None, None]
        d = torch.nn.functional.normalize(X * cam[n, 1] + Y * cam[n, 2] + cam[n, 3], dim=-1).reshape(-1, 3).contiguous()
        o = cam[n, 0].expand(H * W, 3).contiguous()
        t, tri, uv = orc.closest(mesh['v_pos'], mesh['t_pos_idx'], o, d, n_threads=NT)
        got_tri = rast[n, ..., 3].reshape(-1).long() - 1
        same = got_tri == tri.long()
        assert same.float().mean().item() > 0.995                   # silhouette pixels may flip with the ray's last bit
        hit = same & (tri >= 0)
        u_ref = 1.0 - uv[:, 0] - uv[:, 1]                           # nvdiffrast's u = weight of vertex 0
        assert_close(rast[n, ..., 0].reshape(-1)[hit], u_ref[hit], 2e-4, floor=1.0)
        assert_close(rast[n, ..., 1].reshape(-1)[hit], uv[:, 0][hit], 2e-4, floor=1.0)

    # (2) interpolation / depth: render.py:208-234 restated in torch on the kernel's own rast and rast_db
    v_pos_clip = torch.matmul(torch.nn.functional.pad(mesh['v_pos'], (0, 1), value=1.0)[None], mvp.transpose(1, 2))   # render/renderutils xfm_points
    ref = rl.gbuffer_from_rast({k: mesh[k] for k in MESH_KEYS}, rast, rast_db, v_pos_clip)
    for k in ('gb_pos', 'gb_geometric_normal', 'gb_normal', 'gb_tangent', 'gb_texc'):
        assert_close(gb[k], ref[k], 1e-5, floor=1.0, what=k)
    assert_close(gb['gb_texc_deriv'], ref['gb_texc_deriv'], 1e-4, floor=max(ref['gb_texc_deriv'].abs().max().item(), 1e-6), what='gb_texc_deriv')
    assert_close(gb['gb_depth'][..., 0], ref['gb_depth'][..., 0], 1e-5, floor=1.0, what='z/w')
    assert_close(gb['gb_depth'][..., 1], ref['gb_depth'][..., 1], 2e-3, floor=max(ref['gb_depth'][..., 1].abs().max().item(), 1e-9), what='|dz|')
    assert_close(rast[..., 2][covered], ref['gb_depth'][..., 0][covered], 1e-5, floor=1.0, what='rast z/w')
    assert ref['gb_depth'][..., 1][covered].max().item() > 0

    ctx.check()


def test_rast_db_is_the_derivative_of_the_barycentrics(dev):
    """Central differences of u, v across neighbouring pixels of the SAME triangle reproduce the analytic rast_db."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('spot')
    md = {k: mesh[k].to(dev) for k in MESH_KEYS}
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, md['v_pos'], md['t_pos_idx'], 1)
    mvp, cam = _views([2], 8)
    gb = ou.render_gbuffer(ctx, md, mvp.to(dev), cam.to(dev), (640, 640))
    rast, rdb = gb['rast'].cpu(), gb['rast_db'].cpu()
    tid = rast[..., 3]
    for axis, (cu, cv) in ((2, (0, 2)), (1, (1, 3))):                 # X: channels (du/dX, dv/dX); Y: (du/dY, dv/dY)
        lo, mid, hi = [slice(None)] * 3, [slice(None)] * 3, [slice(None)] * 3
        lo[axis], mid[axis], hi[axis] = slice(0, -2), slice(1, -1), slice(2, None)
        lo, mid, hi = tuple(lo), tuple(mid), tuple(hi)
        inner = (tid[mid] > 0) & (tid[mid] == tid[hi]) & (tid[mid] == tid[lo])
        assert inner.float().mean().item() > 0.02
        for ch_b, ch_d in ((0, cu), (1, cv)):
            fd = 0.5 * (rast[..., ch_b][hi] - rast[..., ch_b][lo])
            an = rdb[..., ch_d][mid]
            assert_close(an[inner], fd[inner], 0.02, floor=an[inner].abs().max().item() * 0.05, what='d bary %d / d axis %d' % (ch_b, axis))


def test_trainer_uses_the_gbuffer_kernel(dev):
    """The iteration harness takes mask, attributes, tangents and the (z/w, |dz|) pair from the kernel."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    st = DirectLightingStep('bob', 64, 2, view=[0, 3], device=dev)
    assert st.mask.shape == (2, 64, 64) and set(st.mask.unique().tolist()) <= {0.0, 1.0}
    assert st.gb_depth.shape == (2, 64, 64, 2)
    z = st.gb_depth[..., 0][st.mask > 0]
    assert 0.9 < z.min().item() and z.max().item() < 1.0               # clip-space z/w of a mesh 3 units from a 0.1 / 1000 frustum
    t = st.gb_tangent[st.mask > 0]
    n = st.gb_smooth_nrm[st.mask > 0]
    assert (torch.nn.functional.normalize(t, dim=-1) * torch.nn.functional.normalize(n, dim=-1)).sum(-1).abs().mean().item() < 0.2
    assert torch.isfinite(st.step()).all()
