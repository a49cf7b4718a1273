"""Pin the CPU oracle: against the committed vectors produced by the REFERENCE's own code (tests/golden,
tools/make_golden.py), against oracle/_ref live where this container has the reference, and against a few
analytic properties (white furnace, MIS consistency).  No GPU."""
import math
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as orc, renderutils_ref as rr, scene_cpu
from tests.util import load_npz, checksum, assert_close

sys.path.insert(0, 'tools')
from tools import make_golden as mg  # case tables + input generators (the generator script itself)

NT = min(orc.max_threads(), 16)


# ---------------------------------------------------------------------------------------------- env shade
@pytest.mark.parametrize('case', mg.ENV_CASES, ids=[c[0] for c in mg.ENV_CASES])
def test_env_shade_oracle_vs_reference_vectors(case):
    name, mesh, H, W, n, bsdf, seed, pr, env = case
    gold = load_npz('env_shade_reference.npz')
    inp = mg.env_case_inputs(mesh, H, W, n, pr, env)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    g = torch.Generator().manual_seed(seed)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    assert checksum(*[kw[k] for k in sorted(kw)], dg, sg) == str(gold[name]['inputs_sha256']), 'regenerated inputs differ'
    f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, n_threads=NT, want_vis=True)
    b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=1)
    # (1) the reference built with the same sin/cos/acos/atan2 as ours: every discrete decision identical,
    #     values equal up to powf-vs-multiplication rounding
    gd = gold[name + '/ref_detmath']
    assert np.array_equal(np.packbits(f['vis'].numpy()), gd['vis']), 'visibility bits differ from the reference program'
    for k in ('diff', 'spec'):
        assert_close(f[k], gd[k], 5e-6, what=name + ' ' + k)
    for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad', 'light_grad'):
        assert_close(b[k], gd[k], 1e-3 if k == 'gb_pos_grad' else 5e-4, what=name + ' ' + k)  # powf(x, 1.5|3|4) vs explicit products
    # (2) the reference with libm transcendentals: last-bit differences in a direction may move a sample across a
    #     texel / triangle edge and the GGX peak amplifies them; bounded outliers only
    gl = gold[name + '/ref']
    for k in ('diff', 'spec'):
        assert_close(f[k], gl[k], 1e-4, frac_outliers=0.002, what=name + ' libm ' + k)
    for k in ('gb_normal_grad', 'gb_ks_grad', 'light_grad'):
        assert_close(b[k], gl[k], 1e-3, frac_outliers=0.002, what=name + ' libm ' + k)


@pytest.mark.skipif(not orc.have_ref(), reason='oracle/_ref not built (needs /root/reference)')
def test_env_shade_oracle_vs_reference_live():
    inp = scene_cpu.make_inputs('spot', 48, 48, 4, view=3, env='E1', probe_res=128, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    for bsdf in ('pbr', 'white'):
        a = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=4, rnd_seed=77, n_threads=NT, want_vis=True)
        r = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=4, rnd_seed=77, n_threads=NT, want_vis=True, impl='ref_detmath')
        assert torch.equal(a['vis'], r['vis'])
        assert_close(a['diff'], r['diff'], 5e-6)
        assert_close(a['spec'], r['spec'], 5e-6)


def test_env_shade_white_furnace():
    """Constant environment, no occluder, BSDF 'white': E[diff] = integral of cos/pi * L = L."""
    H = W = 4
    n = 12
    base = torch.full((32, 32, 3), 0.7)
    pdf, cols, rows = orc.light_update_pdf(base)
    mask = torch.ones(1, H, W)
    pos = torch.zeros(1, H, W, 3)
    nrm = torch.nn.functional.normalize(torch.tensor([0.3, 1.0, 0.2]), dim=0).expand(1, H, W, 3).contiguous()
    view = torch.tensor([0.0, 2.0, 0.5]).view(1, 1, 1, 3)
    kd = torch.ones(1, H, W, 3)
    ks = torch.tensor([0.0, 0.5, 0.0]).expand(1, H, W, 3).contiguous()
    verts = torch.tensor([[100.0, -50.0, 0.0], [101.0, -50.0, 0.0], [100.0, -50.0, 1.0]])  # far below the horizon
    tris = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    from nvdiffrecmc_amd.scene import perms_table
    out = orc.env_shade(verts, tris, mask, pos + nrm * 1e-3, pos, nrm, view, kd, ks, base, pdf, rows, cols, perms_table(n),
                        bsdf='white', n_samples_x=n, rnd_seed=5, n_threads=NT)
    assert abs(out['diff'].mean().item() - 0.7) < 0.02
    assert out['spec'].abs().max().item() == 0.0


def test_env_shade_pixel_offset_equals_batch_slice():
    """One-view-per-rank sharding: view r rendered alone with pixel_index_offset = r*H*W draws
    exactly the random streams of slice r of the single-process batch launch (kernel.cu:504)."""
    a = scene_cpu.make_inputs('bob', 24, 24, 2, view=0, probe_res=32, n_threads=NT)
    b = scene_cpu.make_inputs('bob', 24, 24, 2, view=1, probe_res=32, n_threads=NT)
    m = a['mesh']
    ka, kb = scene_cpu.shade_kwargs(a), scene_cpu.shade_kwargs(b)
    batch = dict(ka)
    for k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks'):
        batch[k] = torch.cat([ka[k], kb[k]], 0).contiguous()
    full = orc.env_shade(m['v_pos'], m['t_pos_idx'], **batch, n_samples_x=2, rnd_seed=9, n_threads=NT)
    r0 = orc.env_shade(m['v_pos'], m['t_pos_idx'], **ka, n_samples_x=2, rnd_seed=9, n_threads=NT, pixel_index_offset=0)
    r1 = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kb, n_samples_x=2, rnd_seed=9, n_threads=NT, pixel_index_offset=24 * 24)
    assert torch.equal(full['diff'][0:1], r0['diff']) and torch.equal(full['diff'][1:2], r1['diff'])
    assert torch.equal(full['spec'][1:2], r1['spec'])


# ---------------------------------------------------------------------------------------------- denoiser
@pytest.mark.parametrize('case', mg.DN_CASES, ids=[c[0] for c in mg.DN_CASES])
def test_denoiser_oracle_vs_reference_vectors(case):
    name, N, H, W, sigma, seed = case
    gold = load_npz('denoiser_reference.npz')[name]
    x, col, nrm, zdz, og = mg.denoiser_inputs(N, H, W, seed)
    assert checksum(x, nrm, og) == str(gold['inputs_sha256'])
    assert_close(orc.bilateral_fwd(col, nrm, zdz, sigma, n_threads=NT), gold['out'], 2e-6, floor=1e-4)
    assert_close(orc.bilateral_bwd(col, nrm, zdz, sigma, og, n_threads=NT), gold['col_grad'], 2e-6, floor=1e-4)


def test_denoiser_oracle_vs_authors_torch_formulation():
    """filter_test.py's pure-torch filter (restated in oracle/renderutils_ref.py) incl. autograd through the division."""
    x, col, nrm, zdz, og = mg.denoiser_inputs(1, 20, 18, 4)
    col = col.clone().requires_grad_(True)
    ref = rr.bilateral_denoiser_torch(col, nrm, zdz, 0.9)
    ref.backward(og[..., :3].contiguous())
    o4 = orc.bilateral_fwd(col.detach(), nrm, zdz, 0.9)
    assert_close(o4[..., :3] / o4[..., 3:4], ref.detach(), 1e-5)
    # chain rule through out = rgb / w done by hand: d rgb = g / w ; d w = -sum(g * rgb) / w^2 (not read by the kernel)
    g4 = torch.cat([og[..., :3] / o4[..., 3:4], -(og[..., :3] * o4[..., :3]).sum(-1, keepdim=True) / o4[..., 3:4] ** 2], -1)
    assert_close(orc.bilateral_bwd(col.detach(), nrm, zdz, 0.9, g4.contiguous()), col.grad, 1e-5, floor=1e-4)


# ---------------------------------------------------------------------------------------------- renderutils
def _ru_cases():
    return load_npz('renderutils_reference.npz')


_FN = {
    'fresnel_shlick': rr.fresnel_shlick, 'ndf_ggx': rr.ndf_ggx, 'lambda_ggx': rr.lambda_ggx, 'masking_smith': rr.masking_smith,
    'lambert': rr.lambert, 'frostbite': rr.frostbite, 'pbr_specular': rr.pbr_specular,
    'pbr_bsdf_lambert': lambda *a: rr.pbr_bsdf(*a, bsdf='lambert'), 'pbr_bsdf_frostbite': lambda *a: rr.pbr_bsdf(*a, bsdf='frostbite'),
    'prepare_shading_normal_11': lambda *a: rr.prepare_shading_normal(*a, two_sided_shading=True, opengl=True),
    'prepare_shading_normal_00': lambda *a: rr.prepare_shading_normal(*a, two_sided_shading=False, opengl=False),
    'prepare_shading_normal_bcast': lambda *a: rr.prepare_shading_normal(*a),
    'xfm_points': rr.xfm_points, 'xfm_vectors': rr.xfm_vectors, 'xfm_points_b': rr.xfm_points,
}


@pytest.mark.parametrize('name', sorted(_FN))
def test_renderutils_oracle_vs_reference_module(name):
    c = _ru_cases()[name]
    ins = [torch.from_numpy(c['in%d' % i]).clone().requires_grad_(True) for i in range(int(c['n_in']))]
    out = _FN[name](*ins)
    assert_close(out.detach(), c['out'], 1e-5, floor=1e-4, what=name)
    torch.nn.functional.mse_loss(out, torch.from_numpy(c['target'])).backward()
    for i, t in enumerate(ins):
        assert_close(t.grad, c['grad%d' % i], 2e-5, floor=1e-5, what='%s grad%d' % (name, i))


@pytest.mark.parametrize('loss,tm', [('l1', 'none'), ('l1', 'log_srgb'), ('mse', 'log_srgb'), ('smape', 'none'), ('relmse', 'none'),
                                     ('mse', 'none'), ('n2n', 'none')])
def test_image_loss_oracle_vs_reference_module(loss, tm):
    c = _ru_cases()['image_loss_%s_%s' % (loss, tm)]
    ins = [torch.from_numpy(c['in%d' % i]).clone().requires_grad_(True) for i in range(2)]
    out = rr.image_loss(ins[0], ins[1], loss, tm)
    out.backward()
    assert_close(out.detach(), c['out'], 1e-6)
    for i in range(2):
        assert_close(ins[i].grad, c['grad%d' % i], 1e-5, floor=1e-6)


# ---------------------------------------------------------------------------------------------- light tables
def test_light_tables_oracle_vs_torch_restatement():
    from nvdiffrecmc_amd import scene as sc
    base = sc.env_map('E1', 64)
    pdf, cols, rows = orc.light_update_pdf(base)
    tp, tr, tc = sc.light_tables(base)     # torch transcription of render/light.py:46-59
    assert_close(pdf, tp, 1e-5, floor=1e-9)
    assert_close(cols, tc, 1e-5, floor=1e-6)
    assert_close(rows, tr[:, 0], 1e-5, floor=1e-6)
    assert abs(pdf.sum().item() - 1.0) < 1e-5 and rows[-1].item() == pytest.approx(1.0, abs=1e-6)
    assert (cols[:, 1:] >= cols[:, :-1]).all() and (rows[1:] >= rows[:-1]).all()


@pytest.mark.parametrize('case', mg.LIGHT_CASES, ids=[c[0] for c in mg.LIGHT_CASES])
def test_light_tables_oracle_vs_reference_vectors(case):
    """SURVEY a18 pin: oracle.light_update_pdf against _pdf / cols / rows produced by the REFERENCE's own
    EnvironmentLight.update_pdf (render/light.py:46-59, imported from /root/reference by tools/make_golden.py with its
    CUDA-only imports stubbed); also the repo's torch transcription scene.light_tables, which feeds every other test."""
    from nvdiffrecmc_amd import scene as sc
    name, kind, res = case
    gold = load_npz('light_reference.npz')[name]
    base = mg.light_case_base(kind, res, name)
    assert checksum(base) == str(gold['base_sha256']), 'regenerated probe differs'
    assert np.array_equal(gold['rows'], np.repeat(gold['rows'][:, :1], res, axis=1))   # identical columns: rows[:, 0] is the row CDF
    pdf, cols, rows = orc.light_update_pdf(base)
    # float32 prefix sums in another association than torch.cumsum: 1e-5 relative (floor: the smallest table entries)
    assert_close(pdf, gold['pdf'], 1e-5, floor=1e-9, what=name + ' pdf')
    assert_close(cols, gold['cols'], 1e-5, floor=1e-6, what=name + ' cols')
    assert_close(rows, gold['rows'][:, 0], 1e-5, floor=1e-6, what=name + ' rows')
    tp, tr, tc = sc.light_tables(base)
    assert_close(tp, gold['pdf'], 1e-6, floor=1e-9)
    assert_close(tc, gold['cols'], 1e-6, floor=1e-6)
    assert_close(tr, gold['rows'], 1e-6, floor=1e-6)
    if name.endswith('zero_rows'):
        assert float(cols[5].abs().max()) == 0.0 and float(np.abs(gold['cols'][5]).max()) == 0.0   # light.py:58: 0 / 1, not 0 / 0


# ---------------------------------------------------------------------------------------------- dead samples
@pytest.mark.parametrize('impl', ['oracle', 'ref'])
@pytest.mark.parametrize('mesh,bsdf', [('bob', 'pbr'), ('spot', 'pbr'), ('bob', 'diffuse')])
def test_samples_under_the_shading_horizon_never_matter(impl, mesh, bsdf):
    """The HIP path does not trace shadow rays of samples with dot(n, wi) <= 0.  Justification, checked on the
    reference's own raygen program compiled for the CPU ('ref') and on the restatement: flipping the visibility of
    exactly those samples changes NOTHING -- forward images and all five gradients are bit-identical (the Lambert term is
    max(.,0) = 0 and the GGX lobe fails its front-facing gate, bsdf.h:121,165, forward and backward)."""
    if impl == 'ref' and not orc.have_ref():
        pytest.skip('oracle/_ref not built (no /root/reference on this machine)')
    H = W = 40
    n = 4
    S = n * n
    inp = scene_cpu.make_inputs(mesh, H, W, n, view=4, probe_res=64, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    g = torch.Generator().manual_seed(3)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    common = dict(bsdf=bsdf, n_samples_x=n, rnd_seed=5, n_threads=1)
    base = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, **common, want_vis=True, want_dbg=True)
    dirs = base['dbg'][..., 0:3].view(1, H, W, 2 * S, 3)
    cosn = (dirs * kw['gb_normal'][:, :, :, None, :]).sum(-1).view(H * W, 2 * S)
    dead = cosn < -1e-6                                    # surely under the horizon (margin: torch sums in another order)
    covered = (kw['mask'].view(-1) > 0)
    frac = dead[covered].float().mean().item()
    assert 0.1 < frac < 0.5, frac                          # a quarter of the samples, give or take
    vis_a = base['vis'].clone()
    vis_b = torch.where(dead, 1 - vis_a, vis_a).contiguous()
    assert (vis_a != vis_b)[covered].any()
    outs = []
    for vis in (vis_a, vis_b):
        f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, **common, vis_in=vis, impl=impl)
        b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, **common, vis_in=vis, diff_grad=dg, spec_grad=sg, impl=impl)
        outs.append({**{k: f[k] for k in ('diff', 'spec')},
                     **{k: b[k] for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad', 'light_grad')}})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), '%s: %s depends on the visibility of a sample under the horizon' % (impl, k)
    assert outs[0]['diff'].abs().sum() > 0


@pytest.mark.parametrize('impl', ['oracle', 'ref'])
def test_filter_output_is_empty_where_the_centre_normal_is_zero(impl):
    """Premise of the HIP filter's background early-out, on the reference's own kernel compiled for the CPU: a pixel whose
    normal is exactly zero gets (0, 0, 0, 1e-4) forward and a zero colour gradient, whatever its neighbourhood holds."""
    if impl == 'ref' and not orc.have_ref():
        pytest.skip('oracle/_ref not built (no /root/reference on this machine)')
    x, col, nrm, zdz, og = mg.denoiser_inputs(1, 40, 48, 9)
    nrm = nrm.clone()
    nrm[:, 10:20, 5:30] = 0.0
    nrm[:, 33, 40] = 0.0
    zdz = zdz.contiguous()
    empty = (nrm == 0).all(-1)
    out = orc.bilateral_fwd(col, nrm, zdz, 2.0, n_threads=NT, impl=impl)
    grad = orc.bilateral_bwd(col, nrm, zdz, 2.0, og, n_threads=NT, impl=impl)
    assert torch.equal(out[empty][:, :3], torch.zeros(int(empty.sum()), 3))
    assert torch.equal(out[empty][:, 3], torch.full((int(empty.sum()),), 1e-4))
    assert torch.equal(grad[empty], torch.zeros(int(empty.sum()), 3))
    assert out[~empty][:, 3].min().item() > 1e-4


# ---------------------------------------------------------------------------------------------- mesh frame (round 4)
def test_mesh_frame_oracle_vs_reference_module():
    """oracle/render_layer_ref.py auto_normals + compute_tangents and torch autograd through them against the reference's own
    render/mesh.py:150-219 (tests/golden/mesh_reference.npz, generated by tools/make_golden.py with the CUDA-only imports stubbed)."""
    from oracle import render_layer_ref as rl
    gold = load_npz('mesh_reference.npz')['spot700']
    c = mg.mesh_case()
    assert checksum(c['v_pos'], c['t_pos_idx'], c['v_tex'], c['t_tex_idx'], c['g_nrm'], c['g_tng']) == str(gold['v_pos_sha256']), 'regenerated case differs'
    for dt, tol in ((torch.float32, 1e-6), (torch.float64, 2e-6)):
        v_pos = c['v_pos'].detach().clone().to(dt).requires_grad_(True)
        vn = rl.auto_normals(v_pos, c['t_pos_idx'])
        vt = rl.compute_tangents(v_pos, vn, c['v_tex'], c['t_pos_idx'], c['t_tex_idx'])
        ((vn * c['g_nrm'].to(dt)).sum() + (vt * c['g_tng'].to(dt)).sum()).backward()
        assert_close(vn, gold['v_nrm'], tol, floor=1.0, what='v_nrm')
        assert_close(vt, gold['v_tng'], tol * 10, floor=1.0, what='v_tng')
        assert_close(v_pos.grad, gold['v_pos_grad'], 2e-4, floor=float(np.abs(gold['v_pos_grad']).max()) * 0.01, what='v_pos_grad')


# ---------------------------------------------------------------------------------------------- the literal torch-CPU baseline
def test_torch_bruteforce_shadow_path_matches_the_oracle():
    """BASELINE configs[0]'s "brute-force PyTorch ray-triangle shadow test on CPU" (oracle/torch_baseline.py, timed by bench.py as
    cpu_baseline_torch): its visibility equals the oracle's bit-defined predicate on every CLEAR ray (no (ray, triangle) pair within
    1e-5 of a decision boundary), nearly all rays are clear, and the images / gradients it drives equal the oracle's own."""
    from oracle import scene_cpu, torch_baseline as tb
    n = 2
    inp = scene_cpu.make_inputs('bob', 64, 64, n, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    g = torch.Generator().manual_seed(0)
    dg, sg = torch.rand(1, 64, 64, 3, generator=g), torch.rand(1, 64, 64, 3, generator=g)
    f, b, t = tb.direct_lighting_torch_shadow(m, kw, n, diff_grad=dg, spec_grad=sg, n_threads=NT, want=True)
    ref_f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=0, n_threads=NT, want_vis=True)
    ref_b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=0, n_threads=NT, diff_grad=dg, spec_grad=sg)
    cov = kw['mask'].reshape(-1) > 0
    assert t['covered'] == int(cov.sum()) == ref_f['covered'] and t['rays_per_pass'] == 2 * n * n * t['covered']
    ro, rd = t['rays']
    vis, margin = tb.shadow_rays_bruteforce_torch(m['v_pos'], m['t_pos_idx'], ro, rd, want_margin=True)
    ov = orc.visibility(m['v_pos'], m['t_pos_idx'], ro, rd, n_threads=NT)
    clear = margin > 1e-5
    assert clear.float().mean().item() > 0.98
    assert torch.equal(vis[clear], ov[clear])                                    # bit for bit on the clear rays
    assert (vis != ov).float().mean().item() < 2e-3
    if torch.equal(t['vis'][cov], ref_f['vis'][cov]):                            # (the usual case: no ray on a boundary at all)
        assert torch.equal(f['diff'], ref_f['diff']) and torch.equal(f['spec'], ref_f['spec'])
        assert torch.equal(b['gb_kd_grad'], ref_b['gb_kd_grad']) and torch.equal(b['gb_normal_grad'], ref_b['gb_normal_grad'])
        assert_close(b['light_grad'], ref_b['light_grad'], 1e-5, floor=float(ref_b['light_grad'].abs().max()))     # summed over OpenMP threads
