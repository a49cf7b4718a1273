"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties (the oracle's
brute force cannot run 512x512 x 64..256 spp in seconds) plus an oracle comparison on a sparse pixel subset."""
import pytest
import torch

from oracle import oracle as orc
from nvdiffrecmc_amd import scene as sc
from tests.util import assert_close, vis_by_sample_from_stratum_bits

pytestmark = pytest.mark.gpu
NT = orc.max_threads()


def _gpu_scene(mesh_name, res, n, dev, view=0, env='E1', subdiv=0):
    """Full-size inputs built on the GPU with the product's own G-buffer producer (trace_closest)."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    if subdiv:
        mesh['v_pos'], mesh['t_pos_idx'] = sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], subdiv)
        mesh['v_nrm'] = sc.auto_normals(mesh['v_pos'], mesh['t_pos_idx'])
    md = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in mesh.items()}
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, md['v_pos'], md['t_pos_idx'], 1)
    mv, _, campos = sc.camera(view, 8)
    ro, rd = sc.primary_rays(mv, res, res)
    t, tri, uv = ou.trace_closest(ctx, ro.to(dev), rd.to(dev))
    gb = sc.gbuffer_from_hits(md, t.view(res, res), tri.view(res, res), uv.view(res, res, 2), ro, rd,
                              kd_mode='flat' if subdiv else 'texture')
    base = sc.env_map(env, 256).to(dev)
    pdf, rows, cols = sc.light_tables(base)
    kw = {'mask': gb['mask'], 'gb_pos': gb['gb_pos'], 'gb_normal': gb['gb_normal'],
          'gb_view_pos': campos.to(dev)[None, None, None, :].contiguous(), 'gb_kd': gb['kd'], 'gb_ks': gb['ks'],
          'light': base, 'pdf': pdf, 'rows': rows[:, 0], 'cols': cols}
    kw['ro'] = (kw['gb_pos'] + kw['gb_normal'] * 0.001).contiguous()
    perms = sc.perms_table(n)
    ou.ops.set_permutation_table(n, perms.to(dev))
    return mesh, ctx, kw, perms


def _shade(ctx, kw, n, seed, bsdf='pbr', shadow_scale=1.0, light=None):
    from nvdiffrecmc_amd import optixutils as ou
    return ou.optix_env_shade(ctx, kw['mask'], kw['ro'], kw['gb_pos'], kw['gb_normal'], kw['gb_view_pos'], kw['gb_kd'], kw['gb_ks'],
                              kw['light'] if light is None else light, kw['pdf'], kw['rows'], kw['cols'], BSDF=bsdf,
                              n_samples_x=n, rnd_seed=seed, shadow_scale=shadow_scale)


@pytest.mark.parametrize('mesh_name,n', [('bob', 8), ('spot', 16)])   # configs[1] and configs[2]: 512x512, 64 / 256 spp
def test_fullsize_properties(mesh_name, n, dev):
    res = 512
    mesh, ctx, kw, perms = _gpu_scene(mesh_name, res, n, dev)
    d1, s1 = _shade(ctx, kw, n, 11)
    d2, s2 = _shade(ctx, kw, n, 11)
    assert torch.equal(d1, d2) and torch.equal(s1, s2)                       # deterministic for a fixed seed
    assert torch.isfinite(d1).all() and torch.isfinite(s1).all()
    assert (d1[kw['mask'] <= 0] == 0).all() and (d1 >= 0).all() and (s1 >= 0).all()
    # linearity in the radiance with the sampling tables held fixed: x2 is exact in fp32
    d3, s3 = _shade(ctx, kw, n, 11, light=kw['light'] * 2.0)
    assert torch.equal(d3, d1 * 2.0) and torch.equal(s3, s1 * 2.0)
    # shadows only remove light: unshadowed >= half-shadowed >= shadowed, per pixel and channel
    du, su = _shade(ctx, kw, n, 11, shadow_scale=0.0)
    dh, sh = _shade(ctx, kw, n, 11, shadow_scale=0.5)
    eps = 1e-5 * (1 + du.abs())
    assert (du + eps >= dh).all() and (dh + eps >= d1).all() and (su + 1e-5 * (1 + su.abs()) >= s1).all()
    assert_close(dh, 0.5 * (du + d1), 1e-4, floor=1e-3)                     # V = vis*ss + (1-ss) is affine in ss
    # another seed gives a different but statistically equal image
    d4, _ = _shade(ctx, kw, n, 12)
    assert not torch.equal(d4, d1)
    m = kw['mask'] > 0
    assert abs(d4[m].mean().item() - d1[m].mean().item()) < 0.01 * d1[m].mean().item()


def test_fullsize_sparse_subset_vs_oracle(dev):
    """bob 512x512, 64 spp: every 24th pixel in x and y (the same launch geometry, linear pixel indices and seeds as the
    full frame) against the oracle's brute force, forward and backward."""
    res, n, seed = 512, 8, 3
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev)
    sub = torch.zeros_like(kw['mask'])
    sub[:, 7::24, 5::24] = kw['mask'][:, 7::24, 5::24]
    kws = dict(kw, mask=sub)
    g = torch.Generator().manual_seed(1)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kws[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    kq = dict(kws, **leaves)
    d, s = _shade(ctx, kq, n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    assert 50 < f['covered'] < 200
    assert_close(d, f['diff'], 2e-6)
    assert_close(s, f['spec'], 2e-6)
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-6), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 1e-4, floor=1e-3 * b['light_grad'].abs().max().item())
    # the same benchmark-sized launch against the REFERENCE's own raygen program (oracle/_ref: kernel.cu compiled for the CPU,
    # travels to the GPU box prebuilt): with this library's transcendentals tightly, with libm on >= 99.8 % of the values
    if orc.have_ref():
        for impl, rtol, outl in (('ref_detmath', 5e-6, 0.0), ('ref', 1e-4, 0.002)):
            rf = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT, impl=impl)
            rb = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg,
                               n_threads=NT, impl=impl)
            assert_close(d, rf['diff'], rtol, frac_outliers=outl, what=impl + ' diff')
            assert_close(s, rf['spec'], rtol, frac_outliers=outl, what=impl + ' spec')
            assert_close(leaves['gb_normal'].grad, rb['gb_normal_grad'], 1e-3, floor=1e-3 * rb['gb_normal_grad'].abs().max().item(),
                         frac_outliers=outl, what=impl + ' gb_normal_grad')
            assert_close(leaves['light'].grad, rb['light_grad'], 1e-3, floor=1e-3 * rb['light_grad'].abs().max().item(),
                         frac_outliers=outl, what=impl + ' light_grad')


def test_whole_frame_vs_oracle_with_gpu_visibility(dev):
    """BASELINE configs[1], ONE WHOLE VIEW: bob 512x512, n_samples_x = 8, EVERY covered pixel (~56 k pixels, 7.2 M samples) against
    the oracle, forward and all five gradients.  Brute-force visibility of 7.2 M rays is out of reach of the CPU, so the oracle
    is handed the visibility the GPU traced (the kernels' own bit planes, converted from stratum order to sample order through
    the pixel's permutation rows); everything else -- sample generation, pdfs, texels, BSDF, adjoints -- is the oracle's own.
    The visibility itself is pinned separately: bit-exact vs brute force on random, grazing and real shadow rays (test_gpu_bvh.py)
    and on the sparse subset of this very frame (test_fullsize_sparse_subset_vs_oracle)."""
    from nvdiffrecmc_amd import optixutils as ou
    res, n, seed = 512, 8, 3
    S = n * n
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev)
    g = torch.Generator().manual_seed(1)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kw[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    kq = dict(kw, **leaves)
    ctx.cache_visibility = False                                     # the backward pass re-traces, as the benchmark does
    d, s = _shade(ctx, kq, n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    d2, s2, bits = ou.ops.env_shade_forward_with_bits(ctx, kw['mask'], kw['ro'], kw['gb_pos'], kw['gb_normal'], kw['gb_view_pos'], kw['gb_kd'],
                                                     kw['gb_ks'], kw['light'], kw['pdf'], kw['rows'], kw['cols'], n_samples_x=n, rnd_seed=seed)
    assert torch.equal(d2, d.detach()) and torch.equal(s2, s.detach())
    vis = torch.from_numpy(vis_by_sample_from_stratum_bits(bits.cpu().numpy(), perms.numpy(), seed, S))
    cpu = {k: v.detach().cpu().contiguous() for k, v in kw.items()}
    covered = int((cpu['mask'] > 0).sum())
    occluded = 1.0 - vis.view(-1, 2 * S)[(cpu['mask'] > 0).view(-1)].float().mean().item()
    assert covered > 40000 and 0.02 < occluded < 0.6, (covered, occluded)
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT, vis_in=vis)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg,
                      n_threads=NT, vis_in=vis)
    assert f['covered'] == covered
    assert_close(d, f['diff'], 2e-6, what='diff')
    assert_close(s, f['spec'], 2e-6, what='spec')
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-6), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 2e-4, floor=1e-3 * b['light_grad'].abs().max().item(), what='light')
    ctx.check()


@pytest.mark.parametrize('res', [128, 512], ids=['128', 'benchmarked_8x512'])
def test_batched_views_equal_single_view_launches(res, dev):
    """The benchmarked launch shape -- N = 8 views in ONE launch (configs/bob.json:8), at 128^2 and at the 8 x 512^2 the bench line is
    quoted on -- against eight one-view launches whose pixel index is offset by view * H * W (the data-parallel split, kernel.cu:504):
    images and per-pixel gradients bit for bit, the light gradient (a sum over all views) up to addition order.  Needs no oracle."""
    from nvdiffrecmc_amd import optixutils as ou
    n, seed, nv = 8, 6, 8
    views = [_gpu_scene('bob', res, n, dev, view=v) for v in range(nv)]
    mesh, ctx = views[0][0], views[0][1]
    cat = {k: torch.cat([v[2][k] for v in views], 0).contiguous() for k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks')}
    shared = {k: views[0][2][k] for k in ('light', 'pdf', 'rows', 'cols')}
    g = torch.Generator().manual_seed(2)
    dg, sg = torch.rand(nv, res, res, 3, generator=g).to(dev), torch.rand(nv, res, res, 3, generator=g).to(dev)
    names = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')

    def run(kw, dgv, sgv, offset):
        leaves = {k: kw[k].clone().requires_grad_(True) for k in names}
        ctx.pixel_index_offset = offset
        ctx.cache_visibility = False
        d, s = _shade(ctx, dict(kw, **leaves), n, seed)
        ((d * dgv).sum() + (s * sgv).sum()).backward()
        ctx.pixel_index_offset = None
        return d.detach(), s.detach(), {k: leaves[k].grad for k in names}

    D, Sp, G = run(dict(cat, **shared), dg, sg, 0)
    light_sum = torch.zeros_like(G['light'])
    for v in range(nv):
        kw1 = dict({k: cat[k][v:v + 1].contiguous() for k in cat}, **shared)
        d1, s1, g1 = run(kw1, dg[v:v + 1], sg[v:v + 1], v * res * res)
        assert torch.equal(d1[0], D[v]) and torch.equal(s1[0], Sp[v]), 'view %d' % v
        for k in names[:4]:
            assert torch.equal(g1[k][0], G[k][v]), 'view %d %s' % (v, k)
        light_sum += g1['light']
    assert_close(G['light'], light_sum, 1e-4, floor=1e-3 * light_sum.abs().max().item())
    ctx.check()


@pytest.mark.parametrize('cache_vis', [False, True], ids=['retrace', 'cached_visibility'])
def test_config3_spot_256spp_sparse_subset_vs_oracle(cache_vis, dev):
    """BASELINE configs[2]: spot, 512x512, n_samples_x = 16 (S = 256: four full rounds of 64 lanes per pixel, eight words
    of cached visibility bits per plane) on a sparse pixel subset against the oracle's brute force, forward and backward,
    with the re-tracing backward and with the forward's visibility bits replayed (env_shade.hip stage 3)."""
    from nvdiffrecmc_amd import optixutils as ou
    res, n, seed = 512, 16, 5
    mesh, ctx, kw, perms = _gpu_scene('spot', res, n, dev)
    sub = torch.zeros_like(kw['mask'])
    sub[:, 11::29, 3::31] = kw['mask'][:, 11::29, 3::31]
    kws = dict(kw, mask=sub)
    g = torch.Generator().manual_seed(2)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kws[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    kq = dict(kws, **leaves)
    ctx.cache_visibility = cache_vis
    d, s = _shade(ctx, kq, n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    assert 30 < f['covered'] < 200
    assert_close(d, f['diff'], 4e-6)            # 512 additions per pixel instead of 128: twice the summation-order slack
    assert_close(s, f['spec'], 4e-6)
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-3), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 1e-4, floor=1e-3 * b['light_grad'].abs().max().item())
    ctx.check()


def test_chunked_ray_stream_is_bit_identical(dev):
    """The ray stream holds one chunk of covered pixels (OptiXContext.set_stream_budget): a launch cut into many chunks gives
    the bits of the one-chunk launch, forward and backward (per-pixel gradients exactly, the light gradient up to add order)."""
    from nvdiffrecmc_amd import optixutils as ou
    res, n, seed = 256, 4, 9
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev)
    g = torch.Generator().manual_seed(3)
    dg, sg = torch.rand(1, res, res, 3, generator=g).to(dev), torch.rand(1, res, res, 3, generator=g).to(dev)

    def run(cache_vis):
        leaves = {k: kw[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
        ctx.cache_visibility = cache_vis
        d, s = _shade(ctx, dict(kw, **leaves), n, seed)
        ((d * dg).sum() + (s * sg).sum()).backward()
        return [d.detach(), s.detach()] + [leaves[k].grad for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')]

    whole = run(False)
    ctx.set_stream_budget(1)                               # 1 MB / (32 rays x 25 B + 16 B) -> 1285 pixels per chunk: ~12 non-empty chunks of 51
    for cache_vis in (False, True):
        parts = run(cache_vis)
        for a, b in zip(whole[:6], parts[:6]):
            assert torch.equal(a, b)
        assert_close(parts[6], whole[6], 1e-4, floor=1e-3 * whole[6].abs().max().item())
    ctx.set_stream_budget(2048)
    again = run(False)
    for a, b in zip(whole[:6], again[:6]):
        assert torch.equal(a, b)


def test_light_gradient_band_gather_equals_atomics(dev, monkeypatch):
    """The LDS band gather of the light gradient (no global atomics) against the reference's formulation (three atomicAdds
    per sample, kernel.cu:203-211; NVDR_DEBUG bit 16 selects it): same sums up to the order of the additions."""
    from nvdiffrecmc_amd import optixutils as ou
    res, n, seed = 192, 8, 4
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev)
    g = torch.Generator().manual_seed(5)
    dg, sg = torch.rand(1, res, res, 3, generator=g).to(dev), torch.rand(1, res, res, 3, generator=g).to(dev)

    def light_grad(c):
        light = kw['light'].clone().requires_grad_(True)
        d, s = _shade(c, dict(kw, light=light), n, seed)
        ((d * dg).sum() + (s * sg).sum()).backward()
        return light.grad

    gathered = light_grad(ctx)
    monkeypatch.setenv('NVDR_DEBUG', '16')
    ctx2 = ou.OptiXContext()
    monkeypatch.delenv('NVDR_DEBUG')
    ou.optix_build_bvh(ctx2, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 1)
    atomics = light_grad(ctx2)
    assert atomics.abs().max().item() > 0
    assert_close(gathered, atomics, 1e-4, floor=1e-3 * atomics.abs().max().item())
    # a second backward through the same graph regenerates the stream the first one consumed
    light = kw['light'].clone().requires_grad_(True)
    d, s = _shade(ctx, dict(kw, light=light), n, seed)
    loss = (d * dg).sum() + (s * sg).sum()
    loss.backward(retain_graph=True)
    g1 = light.grad.clone()
    light.grad = None
    loss.backward()
    assert_close(light.grad, g1, 1e-4, floor=1e-3 * g1.abs().max().item())


def test_config5_dmtet_sized_256spp_sparse_subset_vs_oracle(dev):
    """BASELINE configs[4] stand-in (`bench.py --config hotdog512x256`): 512x512, n_samples_x = 16 on a mesh of the size DMTet extracts
    from a 128^3 grid (bob subdivided twice, 171 008 triangles), sparse pixel subset against the oracle's brute force over every
    triangle, forward and all five gradients."""
    res, n, seed = 512, 16, 9
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev, view=3, subdiv=2)
    assert ctx.bvh_info()['n_tris'] == 171008
    sub = torch.zeros_like(kw['mask'])
    sub[:, 7::37, 5::41] = kw['mask'][:, 7::37, 5::41]
    kws = dict(kw, mask=sub)
    g = torch.Generator().manual_seed(4)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kws[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    d, s = _shade(ctx, dict(kws, **leaves), n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    assert 15 < f['covered'] < 120
    assert_close(d, f['diff'], 4e-6)
    assert_close(s, f['spec'], 4e-6)
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-3), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 1e-4, floor=1e-3 * b['light_grad'].abs().max().item())
    ctx.check()


def test_config4_684k_mesh_800_chunked_sparse_subset_vs_oracle(dev):
    """BASELINE configs[3] stand-in at the benchmarked shape (`bench.py --config dmtet800`): 800x800, n_samples_x = 8 on the 684 032
    triangle mesh, the launch cut into several chunks of the ray stream as the 8-view benchmark launch is; a sparse pixel subset
    against the oracle's brute force over every triangle, forward and all five gradients."""
    res, n, seed = 800, 8, 12
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev, view=2, subdiv=3)
    assert ctx.bvh_info()['n_tris'] == 684032
    sub = torch.zeros_like(kw['mask'])
    sub[:, 5::13, 9::17] = kw['mask'][:, 5::13, 9::17]
    kws = dict(kw, mask=sub)
    ctx.set_stream_budget(1)                   # 1 MB / (128 rays x 25 B + 16 B): 326 pixels per chunk
    g = torch.Generator().manual_seed(8)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kws[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    d, s = _shade(ctx, dict(kws, **leaves), n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    assert 400 < f['covered'] < 1500           # at least two chunks
    assert_close(d, f['diff'], 2e-6)
    assert_close(s, f['spec'], 2e-6)
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-3), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 1e-4, floor=1e-3 * b['light_grad'].abs().max().item())
    ctx.check()


def test_config4_dmtet_extraction_800_sparse_subset_vs_oracle(dev):
    """BASELINE configs[3] on a mesh of the kind DMTet produces (`bench.py --config dmtet64_800`: marching tets over the reference's tet
    grid tiled to the 128^3 class, a seeded rough surface with floaters, 140 114 irregular triangles; tools/make_dmtet_mesh.py): 800x800,
    n_samples_x = 8, a sparse pixel subset against the oracle's brute force over every triangle, forward and all five gradients, the launch
    cut into chunks of the ray stream."""
    res, n, seed = 800, 8, 21
    mesh, ctx, kw, perms = _gpu_scene('dmtet64_mid', res, n, dev, view=6)
    assert ctx.bvh_info()['n_tris'] == 140114
    sub = torch.zeros_like(kw['mask'])
    sub[:, 3::13, 5::17] = kw['mask'][:, 3::13, 5::17]
    kws = dict(kw, mask=sub)
    ctx.set_stream_budget(1)
    g = torch.Generator().manual_seed(9)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    leaves = {k: kws[k].clone().requires_grad_(True) for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')}
    d, s = _shade(ctx, dict(kws, **leaves), n, seed)
    ((d * dg.to(dev)).sum() + (s * sg.to(dev)).sum()).backward()
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    assert 400 < f['covered'] < 3000
    assert_close(d, f['diff'], 2e-6)
    assert_close(s, f['spec'], 2e-6)
    for k in ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks'):
        ref = b[k + '_grad']
        assert_close(leaves[k].grad, ref, 2e-4, floor=1e-3 * max(ref.abs().max().item(), 1e-3), what=k)
    assert_close(leaves['light'].grad, b['light_grad'], 1e-4, floor=1e-3 * b['light_grad'].abs().max().item())
    ctx.check()


def test_dmtet_sized_mesh_800(dev):
    """configs[3] stand-in: 800x800, n_samples_x = 8 on a 171k-triangle mesh (bob subdivided twice): finite, deterministic,
    and identical visibility-driven result after a refit to the same vertices."""
    from nvdiffrecmc_amd import optixutils as ou
    res, n = 800, 8
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev, view=5, subdiv=2)
    assert ctx.bvh_info()['n_tris'] == 171008
    d1, s1 = _shade(ctx, kw, n, 2)
    assert torch.isfinite(d1).all() and d1.sum().item() > 0
    ou.optix_build_bvh(ctx, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 0)
    d2, s2 = _shade(ctx, kw, n, 2)
    assert torch.equal(d1, d2) and torch.equal(s1, s2)


def test_training_step_on_a_batch_of_views(dev):
    """The harness renders a batch of views per iteration like the reference (configs/bob.json:8); a two-view batch is the
    concatenation of its views (same RNG streams: the pixel index runs over the batch) and still optimises."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    both = DirectLightingStep('bob', 96, 4, view=[1, 5], device=dev, lr=0.03)
    assert both.mask.shape == (2, 96, 96) and both.view_pos.shape == (2, 1, 1, 3)
    one = DirectLightingStep('bob', 96, 4, view=[5], device=dev, lr=0.03, pixel_index_offset=96 * 96)
    assert torch.equal(both.target[1], one.target[0])          # second view of the batch == the same view alone, offset seeds
    losses = [both.step().item() for _ in range(20)]
    assert sum(losses[-4:]) / 4 < 0.85 * sum(losses[:3]) / 3


def test_training_step_reduces_loss(dev):
    """The iteration the benchmark times (trainer.py) actually optimises: the image loss goes down."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    st = DirectLightingStep('bob', 128, 4, view=1, device=dev, lr=0.03)
    losses = [st.step().item() for _ in range(25)]
    assert all(map(lambda v: v == v, losses))
    assert sum(losses[-5:]) / 5 < 0.8 * sum(losses[:3]) / 3
    # cached-visibility backward gives the gradients of the re-tracing backward
    st.retrace_backward = True
    st.seed = 100
    st.forward_backward()
    g1 = [p.grad.clone() for p in st.params]
    st.retrace_backward = False
    st.seed = 100
    st.forward_backward()
    for a, b in zip(g1, [p.grad for p in st.params]):
        assert_close(b, a, 1e-4, floor=1e-3 * a.abs().max().item())
    # the fused composite / direct filter calls give the gradients of the reference's own sequence of torch calls
    st.retrace_backward = True
    st.fused = False
    st.seed = 100
    l0 = st.forward_backward().item()
    g0 = [p.grad.clone() for p in st.params]
    st.fused = True
    st.seed = 100
    l1 = st.forward_backward().item()
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    for a, b in zip(g0, [p.grad for p in st.params]):
        # (texels of the 1024^2 textures that a single pixel of this 128^2 view feeds: fp32 in another association, 17 of 3.1 M
        # elements between 1e-4 and 3.4e-4 of the floor)
        assert_close(b, a, 1e-4, floor=1e-3 * a.abs().max().item(), frac_outliers=1e-4)


def test_hip_graph_iteration_matches_eager(dev):
    """The iteration captured in a HIP graph (trainer use_graph=True: one submit per iteration) optimises like the eager one:
    same losses step by step -- possible because the shade() seed counter lives in device memory and nothing on the path
    synchronises the host or allocates after warm-up."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    eager = DirectLightingStep('bob', 96, 4, view=[2], device=dev, lr=0.02)
    graph = DirectLightingStep('bob', 96, 4, view=[2], device=dev, lr=0.02, use_graph=True)
    le = [eager.step().item() for _ in range(10)]
    lg = [graph.step().item() for _ in range(10)]
    assert graph._graphs is not None and eager._graphs is None
    assert eager.seed == graph.seed == 11                      # one shade() per iteration + the target render
    for a, b in zip(le, lg):
        assert abs(a - b) < 2e-3 * abs(a), (le, lg)
    assert lg[-1] < lg[0]
    graph.ctx.check()


@pytest.mark.parametrize('probe,n,res,mode', [
    ((256, 512), 4, 96, None),      # 16 bands (the most the records path takes), 4 pixels per wavefront round
    ((64, 64), 2, 64, None),        # one band, 16 pixels per round
    ((256, 256), 16, 48, None),     # S = 256: four rounds per pixel, a group of 512 slots = four blocks
    ((256, 256), 8, 128, '0'),      # the work split of large launches (every workgroup walks all bands) on a small one
    ((256, 256), 8, 128, '1'),      # ... and the per-band split
    ((512, 512), 4, 64, None),      # 32 bands: too many for the records path, falls back to the atomics
])
def test_light_gradient_records_vs_atomics_over_shapes(probe, n, res, mode, dev, monkeypatch):
    """The band-sorted record blocks + LDS gather against the reference's atomics formulation (NVDR_DEBUG bit 16) over probe shapes
    (1, 8, 16, 32 bands), sample counts (several pixels per wavefront round, several rounds per pixel) and both work splits of the
    gather; the per-pixel gradients, which the same kernel writes, must not care at all."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh, ctx0, kw, perms = _gpu_scene('bob', res, n, dev, view=2)
    g = torch.Generator().manual_seed(9)
    base = (torch.rand(probe[0], probe[1], 3, generator=g) + 0.05).to(dev)
    pdf, rows, cols = sc.light_tables(base)
    kw = dict(kw, light=base, pdf=pdf, rows=rows[:, 0], cols=cols)
    dg, sg = torch.rand(1, res, res, 3, generator=g).to(dev), torch.rand(1, res, res, 3, generator=g).to(dev)

    def grads(c):
        light = kw['light'].clone().requires_grad_(True)
        kd = kw['gb_kd'].clone().requires_grad_(True)
        d, s = _shade(c, dict(kw, light=light, gb_kd=kd), n, 3)
        ((d * dg).sum() + (s * sg).sum()).backward()
        c.check()
        return light.grad, kd.grad

    if mode is not None:
        monkeypatch.setenv('NVDR_LG_MODE', mode)
    ctx = ou.OptiXContext()
    monkeypatch.delenv('NVDR_LG_MODE', raising=False)
    ou.optix_build_bvh(ctx, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 1)
    lg, kg = grads(ctx)
    monkeypatch.setenv('NVDR_DEBUG', '16')
    ctx2 = ou.OptiXContext()
    monkeypatch.delenv('NVDR_DEBUG')
    ou.optix_build_bvh(ctx2, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 1)
    la, ka = grads(ctx2)
    assert la.abs().max().item() > 0
    assert_close(lg, la, 1e-4, floor=1e-3 * la.abs().max().item())
    assert torch.equal(kg, ka)
    lg2, _ = grads(ctx)                                       # the tag array is clean again after a launch
    assert_close(lg2, lg, 1e-4, floor=1e-3 * la.abs().max().item())


@pytest.mark.parametrize('cache_vis', [False, True], ids=['retrace', 'cached_visibility'])
def test_queue_shading_kernels_equal_the_plain_ones(cache_vis, dev, monkeypatch):
    """n_samples_x = 8 (S = 64): the shading kernels that queue the live light samples across pixels (env_shade_queue_kernel, the
    default there) against the plain ones (NVDR_SHADE_QUEUE=0): images bit for bit (per lane the same two addends), per-pixel
    gradients up to the order in which a sample's own terms and a pixel's 64 lanes are added, the light gradient up to the order of the records."""
    n, res = 8, 160
    from nvdiffrecmc_amd import optixutils as ou
    seed, nv = 12, 3
    views = [_gpu_scene('bob', res, n, dev, view=v) for v in range(nv)]
    mesh = views[0][0]
    kw = {k: torch.cat([v[2][k] for v in views], 0).contiguous() for k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks')}
    kw.update({k: views[0][2][k] for k in ('light', 'pdf', 'rows', 'cols')})
    g = torch.Generator().manual_seed(3)
    dg, sg = torch.rand(nv, res, res, 3, generator=g).to(dev), torch.rand(nv, res, res, 3, generator=g).to(dev)
    names = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')

    def run(flag):
        monkeypatch.setenv('NVDR_SHADE_QUEUE', flag)
        ctx = ou.OptiXContext()
        monkeypatch.delenv('NVDR_SHADE_QUEUE')
        ou.optix_build_bvh(ctx, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 1)
        ctx.cache_visibility = cache_vis
        leaves = {k: kw[k].clone().requires_grad_(True) for k in names}
        d, s = _shade(ctx, dict(kw, **leaves), n, seed)
        ((d * dg).sum() + (s * sg).sum()).backward()
        ctx.check()
        return d.detach(), s.detach(), {k: leaves[k].grad for k in names}

    d0, s0, g0 = run('0')
    d1, s1, g1 = run('3')
    assert d0.abs().sum().item() > 0
    assert torch.equal(d0, d1) and torch.equal(s0, s1)
    # (5e-5: since round 6 the BACKWARD queue kernel adds a pixel's 64 per-lane results up in another fixed order -- quarter rows out of LDS -- than the plain
    # kernel's butterfly; with the butterfly in both, 2e-5 held.  One element in 230 400 reached 3e-5.)
    for k in names[:4]:
        assert_close(g1[k], g0[k], 5e-5, floor=1e-5 * g0[k].abs().max().item(), what=k)
    assert_close(g1['light'], g0['light'], 1e-4, floor=1e-3 * g0['light'].abs().max().item())
    # the same launch twice: deterministic
    d2, s2, g2 = run('3')
    assert torch.equal(d1, d2) and all(torch.equal(g1[k], g2[k]) for k in names[:4])



@pytest.mark.parametrize('n', [16, 12], ids=['S256', 'S144'])
@pytest.mark.parametrize('cache_vis', [False, True], ids=['retrace', 'cached_visibility'])
def test_local_queue_shading_kernels_vs_the_plain_ones(n, cache_vis, dev, monkeypatch):
    """n_samples_x = 16 / 12 (S = 256 / 144: four / three rounds of 64 lanes per pixel, the last one of S = 144 partial): the shading
    kernels that pack a pixel's live light samples into full 64-lane passes (env_shade_local_kernel; NVDR_SHADE_QUEUE bit 0: backward,
    the default there; bit 2: forward) against the plain ones (NVDR_SHADE_QUEUE=0).  Same samples, same visibility; a queued sample is summed by another lane, so images and
    gradients agree to the rounding of 2 S float additions (the tolerance both have against the oracle), and the result of a pixel does
    not depend on the launch it is part of (one view alone == the same view inside a batch, bit for bit)."""
    res = 96
    from nvdiffrecmc_amd import optixutils as ou
    seed, nv = 7, 2
    views = [_gpu_scene('bob', res, n, dev, view=v) for v in range(nv)]
    mesh = views[0][0]
    kw = {k: torch.cat([v[2][k] for v in views], 0).contiguous() for k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks')}
    kw.update({k: views[0][2][k] for k in ('light', 'pdf', 'rows', 'cols')})
    g = torch.Generator().manual_seed(3)
    dg, sg = torch.rand(nv, res, res, 3, generator=g).to(dev), torch.rand(nv, res, res, 3, generator=g).to(dev)
    names = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')

    def run(flag, first=0, count=nv, offset=0):
        monkeypatch.setenv('NVDR_SHADE_QUEUE', flag)
        ctx = ou.OptiXContext()
        monkeypatch.delenv('NVDR_SHADE_QUEUE')
        ou.optix_build_bvh(ctx, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), 1)
        ctx.cache_visibility = cache_vis
        ctx.pixel_index_offset = offset
        sl = slice(first, first + count)
        kv = {k: (v[sl].contiguous() if k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks') else v) for k, v in kw.items()}
        leaves = {k: kv[k].clone().requires_grad_(True) for k in names}
        d, s = _shade(ctx, dict(kv, **leaves), n, seed)
        ((d * dg[sl]).sum() + (s * sg[sl]).sum()).backward()
        ctx.check()
        return d.detach(), s.detach(), {k: leaves[k].grad for k in names}

    d0, s0, g0 = run('0')
    d1, s1, g1 = run('7')
    assert d0.abs().sum().item() > 0
    assert_close(d1, d0, 2e-6)
    assert_close(s1, s0, 2e-6)
    for k in names[:4]:        # (sums of large terms of both signs: half of the 2e-4 both kernels are allowed against the oracle)
        assert_close(g1[k], g0[k], 1e-4, floor=1e-5 * g0[k].abs().max().item(), what=k)
    assert_close(g1['light'], g0['light'], 1e-4, floor=1e-3 * g0['light'].abs().max().item())
    # deterministic, and a pixel's result is its own: the second view alone (offset seeds) == the second view of the batch
    d2, s2, g2 = run('7')
    assert torch.equal(d1, d2) and all(torch.equal(g1[k], g2[k]) for k in names[:4])
    d3, s3, g3 = run('7', first=1, count=1, offset=res * res)
    assert torch.equal(d3[0], d1[1]) and torch.equal(s3[0], s1[1])
    for k in names[:4]:
        assert torch.equal(g3[k][0], g1[k][1]), k
