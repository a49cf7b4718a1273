"""The gradient route from the G-buffer to the trained vertices and textures (csrc/mesh.hip, SURVEY 8 f1 second half / row g1):
mesh frame (auto_normals + compute_tangents) forward and adjoint, the adjoint of the attribute interpolation with and without
the barycentric term, the texel lookups -- against torch autograd through oracle/render_layer_ref.py in DOUBLE precision on the
CPU and against the reference's own render/mesh.py (tests/golden/mesh_reference.npz)."""
import numpy as np
import pytest
import torch

from oracle import render_layer_ref as rl
from nvdiffrecmc_amd import scene as sc
from tests.util import assert_close, load_npz
from tools import make_golden as mg

pytestmark = pytest.mark.gpu


def _views(views, n_views, aspect=1.0):
    mvp, cam = [], []
    for k in views:
        mv, p, _ = sc.camera(k, n_views, aspect=aspect)
        mvp.append(p)
        cam.append(sc.camera_rays(mv, aspect=aspect))
    return torch.stack(mvp).contiguous(), torch.stack(cam).contiguous()


def _topo(mesh, dev):
    from nvdiffrecmc_amd.mesh import MeshTopology
    return MeshTopology(mesh['t_pos_idx'].to(dev), mesh['v_pos'].shape[0], mesh['v_tex'].to(dev), mesh['t_tex_idx'].to(dev))


def test_mesh_frame_vs_reference_module_vectors(dev):
    """HIP normals / tangents and their adjoint == the reference's render/mesh.py + autograd (committed vectors)."""
    from nvdiffrecmc_amd.mesh import mesh_frame
    gold = load_npz('mesh_reference.npz')['spot700']
    c = mg.mesh_case()
    topo = _topo(c, dev)
    v_pos = c['v_pos'].to(dev).requires_grad_(True)
    vn, vt = mesh_frame(v_pos, topo)
    ((vn * c['g_nrm'].to(dev)).sum() + (vt * c['g_tng'].to(dev)).sum()).backward()
    assert_close(vn, gold['v_nrm'], 2e-6, floor=1.0, what='v_nrm')
    assert_close(vt, gold['v_tng'], 2e-5, floor=1.0, what='v_tng')
    assert_close(v_pos.grad, gold['v_pos_grad'], 2e-4, floor=float(np.abs(gold['v_pos_grad']).max()) * 0.01, what='v_pos_grad')


@pytest.mark.parametrize('mesh_name', ['bob', 'spot'])
def test_mesh_frame_fwd_bwd_vs_oracle_fp64(mesh_name, dev):
    from nvdiffrecmc_amd.mesh import mesh_frame
    m = sc.load_mesh(mesh_name)
    topo = _topo(m, dev)
    g = torch.Generator().manual_seed(3)
    gn, gt = torch.randn(m['v_pos'].shape, generator=g), torch.randn(m['v_pos'].shape, generator=g)
    v_pos = m['v_pos'].to(dev).requires_grad_(True)
    vn, vt = mesh_frame(v_pos, topo)
    ((vn * gn.to(dev)).sum() + (vt * gt.to(dev)).sum()).backward()
    # bit-reproducible: a second evaluation gives the same bits (gathers in a fixed order, no atomics)
    vn2, vt2 = mesh_frame(v_pos.detach(), topo)
    assert torch.equal(vn, vn2) and torch.equal(vt, vt2)
    p64 = m['v_pos'].double().requires_grad_(True)
    rn = rl.auto_normals(p64, m['t_pos_idx'])
    rt = rl.compute_tangents(p64, rn, m['v_tex'], m['t_pos_idx'], m['t_tex_idx'])
    ((rn * gn.double()).sum() + (rt * gt.double()).sum()).backward()
    assert_close(vn, rn, 1e-6, floor=1.0, what='v_nrm')
    assert_close(vt, rt, 2e-5, floor=1.0, what='v_tng')          # two normalisations of a Gram-Schmidt difference in fp32
    scale = float(p64.grad.abs().max())
    assert_close(v_pos.grad, p64.grad, 1e-5, floor=scale, what='v_pos_grad')       # 1e-5 of the gradient's scale
    # normals alone (a mesh without texture coordinates)
    from nvdiffrecmc_amd.mesh import MeshTopology
    topo_n = MeshTopology(m['t_pos_idx'].to(dev), m['v_pos'].shape[0])
    v2 = m['v_pos'].to(dev).requires_grad_(True)
    vn3, vt3 = mesh_frame(v2, topo_n)
    assert torch.equal(vn3, vn.detach()) and float(vt3.abs().max()) == 0.0
    (vn3 * gn.to(dev)).sum().backward()
    q64 = m['v_pos'].double().requires_grad_(True)
    (rl.auto_normals(q64, m['t_pos_idx']) * gn.double()).sum().backward()
    assert_close(v2.grad, q64.grad, 1e-5, floor=float(q64.grad.abs().max()), what='v_pos_grad (normals only)')


@pytest.mark.parametrize('bary', [False, True])
@pytest.mark.parametrize('mesh_name,H,W', [('bob', 96, 96), ('spot', 64, 112)])
def test_interpolate_bwd_vs_autograd_of_the_restatement(mesh_name, H, W, bary, dev):
    """Vertex gradients of the G-buffer attributes: nvdr_interpolate_bwd against torch autograd (fp64, CPU) through the restated
    interpolation on the kernel's own coverage -- barycentrics held constant, and following the vertices along the fixed primary ray."""
    from nvdiffrecmc_amd import optixutils as ou
    from nvdiffrecmc_amd import render as rd
    m = sc.load_mesh(mesh_name)
    topo = _topo(m, dev)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, m['v_pos'].to(dev), topo.t_pos_idx, 1)
    mvp, cam = _views([1, 6], 8, aspect=W / H)
    leaves = [m[k].to(dev).requires_grad_(True) for k in ('v_pos', 'v_nrm', 'v_tng')]
    gb = rd.gbuffer(ctx, *leaves, topo, mvp.to(dev), cam.to(dev), (H, W), bary_grad=bary)
    g = torch.Generator().manual_seed(11)
    ups = {k: torch.randn(2, H, W, 3, generator=g) for k in ('gb_pos', 'gb_geometric_normal', 'gb_normal', 'gb_tangent')}
    sum((gb[k] * ups[k].to(dev)).sum() for k in ups).backward()
    rast = gb['rast'].detach().cpu()
    assert not gb['rast'].requires_grad and not gb['gb_depth'].requires_grad and not gb['gb_texc'].requires_grad
    ref_leaves = [m[k].double().requires_grad_(True) for k in ('v_pos', 'v_nrm', 'v_tng')]
    ref = rl.gbuffer_differentiable(*ref_leaves, m['t_pos_idx'], rast, cam=cam if bary else None)
    for k in ups:
        assert_close(gb[k].detach(), ref[k].detach(), 1e-5, floor=1.0, what=k)
    if bary:    # the barycentrics of the fixed ray, recomputed from the vertices, are the rasteriser's
        ur, vr = rl.ray_barycentrics(m['v_pos'].double(), m['t_pos_idx'], rast, cam)
        cov = rast[..., 3] > 0
        assert_close(ur[..., 0][cov], rast[..., 0][cov], 1e-3, floor=1.0, what='u of the ray')      # fp32 hit of a grazing triangle: 2.4e-4 seen
        assert_close(vr[..., 0][cov], rast[..., 1][cov], 1e-3, floor=1.0, what='v of the ray')
    sum((ref[k] * ups[k].double()).sum() for k in ups).backward()
    for got, want, name in zip(leaves, ref_leaves, ('v_pos', 'v_nrm', 'v_tng')):
        scale = float(want.grad.abs().max())
        assert scale > 0
        # 1e-5 of the gradient's scale: fp32 atomics in an arbitrary order over ~10 pixels per vertex
        assert_close(got.grad, want.grad, 1e-5, floor=scale, what=name + '.grad')
    ctx.check()


def test_barycentric_term_matches_finite_differences(dev):
    """The barycentric term itself: moving ONE vertex changes gb_pos of the pixels of its triangles the way the adjoint says."""
    from nvdiffrecmc_amd import optixutils as ou
    from nvdiffrecmc_amd import render as rd
    m = sc.load_mesh('spot')
    topo = _topo(m, dev)
    H = W = 128
    mvp, cam = _views([3], 8)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, m['v_pos'].to(dev), topo.t_pos_idx, 1)
    leaves = [m[k].to(dev).requires_grad_(True) for k in ('v_pos', 'v_nrm', 'v_tng')]
    gb = rd.gbuffer(ctx, *leaves, topo, mvp.to(dev), cam.to(dev), (H, W))
    rast = gb['rast'].detach().cpu()
    g = torch.Generator().manual_seed(5)
    up = torch.randn(1, H, W, 3, generator=g)
    (gb['gb_normal'] * up.to(dev)).sum().backward()       # gb_normal depends on v_pos ONLY through the barycentrics
    tid = rast[..., 3].long() - 1
    vis_tris = torch.unique(tid[tid >= 0])
    vtx = int(m['t_pos_idx'].long()[vis_tris[len(vis_tris) // 2], 1])
    eps = 1e-4
    fd = torch.zeros(3, dtype=torch.float64)
    for ax in range(3):
        vals = []
        for sgn in (1.0, -1.0):
            p = m['v_pos'].double().clone()
            p[vtx, ax] += sgn * eps
            out = rl.gbuffer_differentiable(p, m['v_nrm'].double(), m['v_tng'].double(), m['t_pos_idx'], rast, cam=cam, ray_values=True)
            vals.append((out['gb_normal'] * up.double()).sum())
        fd[ax] = (vals[0] - vals[1]) / (2 * eps)
    got = leaves[0].grad[vtx].double().cpu()
    assert float(fd.abs().max()) > 1e-3
    assert_close(got, fd, 2e-3, floor=float(fd.abs().max()), what='d gb_normal / d v_pos through the barycentrics')


def test_texture_lookup_fwd_bwd(dev):
    from nvdiffrecmc_amd import render as rd
    g = torch.Generator().manual_seed(2)
    N, H, W = 2, 40, 56
    texc = torch.rand(N, H, W, 2, generator=g) * 1.2 - 0.1               # some coordinates outside [0, 1): clamped
    rast = torch.zeros(N, H, W, 4)
    rast[..., 3] = (torch.rand(N, H, W, generator=g) > 0.4).float() * 7.0
    texs = [torch.rand(R, R, 3, generator=g) for R in (64, 33, 128)]
    ups = [torch.randn(N, H, W, 3, generator=g) for _ in texs]
    leaves = [t.to(dev).requires_grad_(True) for t in texs]
    outs = rd.texture_lookup(leaves, texc.to(dev), rast.to(dev))
    sum((o * u.to(dev)).sum() for o, u in zip(outs, ups)).backward()
    for t, o, u, leaf in zip(texs, outs, ups, leaves):
        t64 = t.double().requires_grad_(True)
        ref = rl.texture_lookup(t64, texc, rast)
        (ref * u.double()).sum().backward()
        assert torch.equal(o.detach().cpu(), ref.detach().float())           # a lookup: bit-exact
        assert_close(leaf.grad, t64.grad, 1e-6, floor=float(t64.grad.abs().max()), what='texture grad')
    # one texture, and the error for too many
    (o1,) = rd.texture_lookup([leaves[0]], texc.to(dev), rast.to(dev))
    assert torch.equal(o1, outs[0])
    with pytest.raises(RuntimeError):
        rd.texture_lookup([leaves[0]] * 5, texc.to(dev), rast.to(dev))


# ---------------------------------------------------------------------------------------------- the iteration with the reference's parameter set
def test_full_material_set_iteration_trains(dev):
    """kd + ks + normal textures and the probe (train.py:171-192) through the whole iteration: every tensor receives a gradient --
    the normal map's through prepare_shading_normal_bwd, which therefore runs inside the iteration -- and the loss goes down."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    st = DirectLightingStep('bob', 128, 4, view=[0, 3], device=dev, tex_res=256, lr=0.03)
    assert st.param_names == ['kd', 'ks', 'normal', 'light']
    assert st.kd_tex.shape == (256, 256, 3) and st.ks_tex.shape == (256, 256, 3) and st.nrm_tex.shape == (256, 256, 3)
    before = [p.detach().clone() for p in st.params]
    losses = [float(st.step().detach()) for _ in range(40)]
    for name, p, p0 in zip(st.param_names, st.params, before):
        assert torch.isfinite(p).all() and float((p.detach() - p0).abs().max()) > 0, name          # every tensor moved
    assert sum(losses[-5:]) < 0.8 * sum(losses[:5]), losses
    st.forward_backward()           # (a step leaves p.grad zeroed where the fused update consumed it: look at a backward pass on its own)
    for name, p in zip(st.param_names, st.params):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, name
    # the clamps of train.py:467-476 hold: ks.x pinned to 0, roughness >= ks_min, unit normals with z >= 0, light >= 0.01
    assert float(st.ks_tex[..., 0].abs().max()) == 0.0 and float(st.ks_tex[..., 1].min()) >= 0.08 - 1e-7
    n = st.nrm_tex.detach()
    assert float(((n * n).sum(-1) - 1).abs().max()) < 1e-5 and float(n[..., 2].min()) >= 0.0
    assert float(st.light.base.min()) >= 0.01 - 1e-8
    # texels no covered pixel looks up keep a zero gradient (the lookup's adjoint scatters, nothing else writes)
    assert float((st.kd_tex.grad.abs().sum(-1) == 0).float().mean()) > 0.3


def test_fused_and_composed_iterations_agree(dev):
    """The fused harness (texture_lookup / shading_frame / pair filter / composite / fused Adam) against the same iteration written
    with torch.optim.Adam and the reference's clamp sequence: same losses step by step (same samples: the seed counter is shared)."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    a = DirectLightingStep('bob', 96, 4, view=[1], device=dev, tex_res=128, fused=True)
    b = DirectLightingStep('bob', 96, 4, view=[1], device=dev, tex_res=128, fused=False)
    for k in range(6):
        la, lb = float(a.step().detach()), float(b.step().detach())
        assert abs(la - lb) <= 2e-5 * max(abs(lb), 1e-3), (k, la, lb)
    for p, q, name in zip(a.params, b.params, a.param_names):
        # an element whose gradient is rounding noise (the light gradient is summed in another order) takes an lr-sized Adam step of
        # either sign: a handful of the 49 152 texels, 245 of the probe's 196 608 values
        assert_close(p.detach(), q.detach(), 2e-4, floor=1.0, frac_outliers=1e-2, what=name)      # (the count varies run to run: atomics)


def test_geometry_unlocked_loss_decreases(dev):
    """DirectLightingStep(optimize_geometry=True): the BVH, the vertex frames and the G-buffer follow the moving vertices every
    iteration and the image loss pulls a perturbed bob back (materials and light held at their true values: lr scale 0)."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    st = DirectLightingStep('bob', 160, 4, view=[0, 2, 5], device=dev, tex_res=512, optimize_geometry=True, perturb_pos=0.004,
                            lr=0.01, lr_pos=1e-4)
    assert st.param_names[-1] == 'v_pos'
    with torch.no_grad():                       # true materials / light: what is left to explain is the geometry
        st.kd_tex.copy_(st.mesh['kd_tex'])
        st.ks_tex.copy_(st.mesh['ks'].view(1, 1, 3).expand_as(st.ks_tex))
        st.light.base.copy_(sc.env_map('E1', 256).to(dev))
    for name in st.param_names[:-1]:
        st.set_lr_scale(name, 0.0)
    v_true = st.mesh['v_pos']
    err0 = float((st.v_pos.detach() - v_true).norm())
    losses = [float(st.step().detach()) for _ in range(48)]
    assert st.v_pos.grad is not None and torch.isfinite(st.v_pos.grad).all() and float(st.v_pos.grad.abs().max()) > 0
    # tools/geo_probe.py: 0.0104 -> 0.0077 over these steps (the noise floor of the loss with the true mesh is 0.0031; there is no
    # Laplacian regulariser and no silhouette term here, and Adam random-walks the vertices that only see Monte-Carlo noise)
    assert sum(losses[-8:]) / 8 < 0.85 * sum(losses[:3]) / 3, losses
    assert float((st.v_pos.detach() - v_true).norm()) != err0          # the vertices moved ...
    info = st.ctx.bvh_info()                                            # ... and the BVH was rebuilt from them
    assert info['n_tris'] == st.mesh['t_pos_idx'].shape[0]
    st.ctx.check()


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hip_graphs'])
def test_refit_policy_trains_like_a_rebuild_every_iteration(graph, dev):
    """rebuild_every=4 (a rebuild, then three refits of its topology to the moved vertices, ...): visibility and primary hits are exact with
    any valid tree, so the iteration computes the same image and the same gradients -- the losses follow the rebuild-every-iteration run step
    by step (up to the order of atomic additions and, rarely, a primary ray grazing an edge picking the neighbouring triangle)."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    runs = {}
    for K in (1, 4):
        st = DirectLightingStep('bob', 128, 4, view=[0, 3], device=dev, tex_res=256, optimize_geometry=True, perturb_pos=0.003, lr=0.01, lr_pos=1e-4,
                                use_graph=graph, rebuild_every=K, subdiv=1)
        runs[K] = [float(st.step().detach()) for _ in range(14)]
        assert (st._graphs is not None) == graph
        if graph and K > 1:
            assert isinstance(st._graphs[0], tuple) and len(st._graphs[0]) == 2          # the iteration captured with a rebuild and with a refit
        st.ctx.check()
    for a, b in zip(runs[1], runs[4]):
        assert abs(a - b) <= 2e-3 * abs(a), (runs[1], runs[4])
