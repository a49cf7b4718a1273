"""The fused env-shade kernel through the public autograd API vs the CPU oracle and the reference vectors."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc, scene_cpu
from tests.util import load_npz, assert_close
from tests.gpu_util import make_ctx, gpu_env_shade
from tools import make_golden as mg

pytestmark = pytest.mark.gpu
NT = orc.max_threads()
GRADS = ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad', 'light_grad')

# forward: every sample's direction, texel and visibility are bit-identical to the oracle; only the order of the
# 2*S additions per pixel differs (butterfly across lanes vs serial loop) -> 2e-6 relative
FWD_RTOL = 2e-6
# backward: same, but the per-pixel gradients are sums of large cancelling terms -> 2e-4 of (|ref| + 1e-3)
BWD_RTOL = 2e-4


@pytest.mark.parametrize('mesh,H,W,n,bsdf,seed', [
    ('bob', 64, 64, 1, 'pbr', 1),          # S = 1: 64 pixels per wavefront
    ('bob', 72, 56, 2, 'pbr', 2),          # ragged extent, 16 pixels per wavefront
    ('bob', 64, 64, 3, 'pbr', 3),          # S = 9: not a power of two (idle lanes)
    ('bob', 48, 48, 8, 'pbr', 4),          # the benchmark setting: one pixel per wavefront
    ('bob', 32, 32, 12, 'pbr', 5),         # S = 144 > 64: strata loop with a partial last round
    ('spot', 64, 64, 4, 'pbr', 6),         # metal (ks.z = 1): exercises the kd gradient
    ('bob', 64, 64, 4, 'diffuse', 7),
    ('bob', 64, 64, 2, 'white', 8),
    ('bob', 32, 32, 16, 'diffuse', 9),     # S = 256: the pixel-local queue kernel on the Lambert arm
    ('spot', 32, 32, 16, 'pbr', 10),       # S = 256, metal
    ('bob', 32, 32, 32, 'pbr', 11),        # S = 1024: the reference's VALIDATION sample count (train.py:263 n_samples = 32), 2 048 rays per pixel,
                                           # sixteen rounds per pixel, the widest rows the light-gradient records take (a pixel's slots = 16 blocks of 128)
    ('bob', 24, 24, 33, 'pbr', 12),        # S = 1089: beyond that limit -- the plain shading kernels with memory-side atomics for the light gradient
])
def test_env_shade_fwd_bwd_vs_oracle(mesh, H, W, n, bsdf, seed, dev):
    inp = scene_cpu.make_inputs(mesh, H, W, n, view=seed % 8, probe_res=128, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    g = torch.Generator().manual_seed(seed)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    ref_f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, n_threads=NT)
    ref_b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=NT)
    ctx = make_ctx(m, dev)
    got = gpu_env_shade(ctx, kw, dev, bsdf, n, seed, dg, sg, cache_vis=False)
    assert ref_f['covered'] > 100 or (n > 32 and ref_f['covered'] > 50)
    # (2 S additions per pixel in another order than the serial loop: the bound grows with their number)
    fwd_rtol = FWD_RTOL * (4.0 if n * n > 256 else 1.0)
    for k in ('diff', 'spec'):
        assert_close(got[k], ref_f[k], fwd_rtol, what=k)
        assert torch.equal(got[k][inp['mask'] <= 0], torch.zeros_like(got[k][inp['mask'] <= 0]))  # zero-initialised outputs
    for k in GRADS:
        assert_close(got[k], ref_b[k], BWD_RTOL, floor=1e-3 * max(1.0, ref_b[k].abs().max().item()), what=k)
    # the visibility bits cached by the forward pass give the same gradients as re-tracing
    got_c = gpu_env_shade(ctx, kw, dev, bsdf, n, seed, dg, sg, cache_vis=True)
    for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad'):
        assert torch.equal(got_c[k], got[k]), k
    assert_close(got_c['light_grad'], got['light_grad'], 1e-4, floor=1e-3 * max(1.0, got['light_grad'].abs().max().item()))


@pytest.mark.parametrize('case', mg.ENV_CASES, ids=[c[0] for c in mg.ENV_CASES])
def test_env_shade_vs_reference_vectors(case, dev):
    """Against outputs of the reference's own raygen program (compiled for the CPU, tests/golden)."""
    name, mesh, H, W, n, bsdf, seed, pr, env = case
    gold = load_npz('env_shade_reference.npz')
    inp = mg.env_case_inputs(mesh, H, W, n, pr, env)
    kw = scene_cpu.shade_kwargs(inp)
    g = torch.Generator().manual_seed(seed)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    ctx = make_ctx(inp['mesh'], dev)
    got = gpu_env_shade(ctx, kw, dev, bsdf, n, seed, dg, sg)
    gd, gl = gold[name + '/ref_detmath'], gold[name + '/ref']
    for k in ('diff', 'spec'):
        assert_close(got[k], gd[k], 5e-6, what=k)                              # same transcendentals: tight
        assert_close(got[k], gl[k], 1e-4, frac_outliers=0.002, what='libm ' + k)  # the reference with libm
    for k in GRADS:
        assert_close(got[k], gd[k], 1e-3, floor=1e-3 * max(1.0, float(np.abs(gd[k]).max())), what=k)


def test_env_shade_shadow_scale_and_batch_and_offset(dev):
    """shadow_scale = 0 ignores visibility; a 2-view batch equals two single views with the rank offset."""
    a = scene_cpu.make_inputs('bob', 40, 40, 4, view=0, probe_res=64, n_threads=NT)
    b = scene_cpu.make_inputs('bob', 40, 40, 4, view=3, probe_res=64, n_threads=NT)
    m = a['mesh']
    ka, kb = scene_cpu.shade_kwargs(a), scene_cpu.shade_kwargs(b)
    ctx = make_ctx(m, dev)
    ref0 = orc.env_shade(m['v_pos'], m['t_pos_idx'], **ka, n_samples_x=4, rnd_seed=1, shadow_scale=0.0, n_threads=NT)
    got0 = gpu_env_shade(ctx, ka, dev, 'pbr', 4, 1, shadow_scale=0.0)
    assert_close(got0['diff'], ref0['diff'], FWD_RTOL)
    refh = orc.env_shade(m['v_pos'], m['t_pos_idx'], **ka, n_samples_x=4, rnd_seed=1, shadow_scale=0.5, n_threads=NT)
    goth = gpu_env_shade(ctx, ka, dev, 'pbr', 4, 1, shadow_scale=0.5)
    assert_close(goth['spec'], refh['spec'], FWD_RTOL)
    batch = dict(ka)
    for k in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks'):
        batch[k] = torch.cat([ka[k], kb[k]], 0).contiguous()
    full = gpu_env_shade(ctx, batch, dev, 'pbr', 4, 9)
    r1 = gpu_env_shade(ctx, kb, dev, 'pbr', 4, 9, offset=40 * 40)
    assert torch.equal(full['diff'][1:2], r1['diff']) and torch.equal(full['spec'][1:2], r1['spec'])


def test_env_shade_api_contract(dev):
    """Shapes, None-gradients, strided inputs and argument errors of the drop-in API (ops.py:78-137)."""
    from nvdiffrecmc_amd import optixutils as ou
    inp = scene_cpu.make_inputs('bob', 32, 32, 2, probe_res=32, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    ctx = make_ctx(inp['mesh'], dev)
    ou.ops.set_permutation_table(2, kw['perms'].to(dev))
    g = {k: v.to(dev) for k, v in kw.items() if k != 'perms'}
    # mask as the strided view rast[..., -1] the reference passes (render.py:113)
    rast = torch.zeros(1, 32, 32, 4, device=dev)
    rast[..., 3] = g['mask']
    for k in ('ro', 'gb_view_pos'):
        g[k].requires_grad_(True)
    g['gb_kd'].requires_grad_(True)
    d1, s1 = ou.optix_env_shade(ctx, rast[..., -1], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                                g['light'], g['pdf'], g['rows'], g['cols'], BSDF='pbr', n_samples_x=2, rnd_seed=4)
    d2, s2 = ou.optix_env_shade(ctx, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                                g['light'], g['pdf'], g['rows'], g['cols'], BSDF='pbr', n_samples_x=2, rnd_seed=4)
    assert d1.shape == (1, 32, 32, 3) and torch.equal(d1, d2) and torch.equal(s1, s2)
    (d1.sum() + s1.sum()).backward()
    assert g['ro'].grad is None and g['gb_view_pos'].grad is None and g['gb_kd'].grad.shape == (1, 32, 32, 3)
    with pytest.raises(ValueError):
        ou.optix_env_shade(ctx, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                           g['light'], g['pdf'], g['rows'], g['cols'], BSDF='phong')
    with pytest.raises(RuntimeError, match='GPU'):
        ou.optix_env_shade(ctx, g['mask'].cpu(), g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                           g['light'], g['pdf'], g['rows'], g['cols'])
    empty = ou.OptiXContext()
    with pytest.raises(RuntimeError, match='BVH'):
        ou.optix_env_shade(empty, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                           g['light'], g['pdf'], g['rows'], g['cols'], n_samples_x=2, rnd_seed=1)
    # rnd_seed=None (FLAGS.decorrelated): runs, different seeds forward/backward, finite
    dn, sn = ou.optix_env_shade(ctx, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                                g['light'], g['pdf'], g['rows'], g['cols'], n_samples_x=2)
    assert torch.isfinite(dn).all() and torch.isfinite(sn).all()
    # fully masked-out image: all zeros, no crash
    z, _ = ou.optix_env_shade(ctx, torch.zeros_like(g['mask']), g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'],
                              g['gb_ks'], g['light'], g['pdf'], g['rows'], g['cols'], n_samples_x=2, rnd_seed=1)
    assert z.abs().max().item() == 0.0


@pytest.mark.parametrize('mesh,n,bsdf', [('bob', 8, 'pbr'), ('spot', 4, 'pbr'), ('bob', 3, 'diffuse')])
def test_dead_samples_are_exactly_zero(mesh, n, bsdf, dev, monkeypatch):
    """Samples under the shading horizon (dot(n, wi) <= 0) are not traced and not shaded; tracing and shading them all
    (NVDR_DEBUG=8, what the reference does) must give bit-identical images and per-pixel gradients."""
    H = W = 64
    inp = scene_cpu.make_inputs(mesh, H, W, n, view=3, probe_res=128, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    g = torch.Generator().manual_seed(11)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    ctx = make_ctx(inp['mesh'], dev)
    fast = gpu_env_shade(ctx, kw, dev, bsdf, n, 9, dg, sg)
    monkeypatch.setenv('NVDR_DEBUG', '8')            # read ONCE when a context is created: this one traces every ray
    ctx_all = make_ctx(inp['mesh'], dev)
    monkeypatch.delenv('NVDR_DEBUG')
    full = gpu_env_shade(ctx_all, kw, dev, bsdf, n, 9, dg, sg)
    for k in ('diff', 'spec', 'gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad'):
        assert torch.equal(fast[k], full[k]), k
    # light gradient: same addends (the skipped ones are zeros), different order
    assert_close(fast['light_grad'], full['light_grad'], 1e-4, floor=1e-3 * max(1.0, full['light_grad'].abs().max().item()))
    # and the traversal really is shorter
    from nvdiffrecmc_amd import optixutils as ou
    d = {k: v.to(dev) for k, v in kw.items()}
    ou.ops.set_permutation_table(n, d['perms'])
    args = (d['mask'], d['ro'], d['gb_pos'], d['gb_normal'], d['gb_view_pos'], d['gb_kd'], d['gb_ks'], d['light'], d['pdf'], d['rows'], d['cols'])
    P, _, _, traced = ou.ops.env_shade_traversal_counts(ctx, *args, BSDF=bsdf, n_samples_x=n, rnd_seed=9)
    P2, _, _, traced_all = ou.ops.env_shade_traversal_counts(ctx_all, *args, BSDF=bsdf, n_samples_x=n, rnd_seed=9)
    assert P == P2 and traced_all == 2 * n * n * P
    assert 0.5 * traced_all < traced < 0.95 * traced_all


def test_backward_after_another_forward_regenerates_the_stream(dev):
    """The backward pass re-uses the forward's ray stream only while it is still the context's most recent one: a second
    forward (other seed) in between, or a BVH rebuild, must make backward regenerate the samples -- same gradients."""
    from nvdiffrecmc_amd import optixutils as ou
    H = W = 48
    n = 4
    inp = scene_cpu.make_inputs('bob', H, W, n, view=2, probe_res=64, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    ctx = make_ctx(m, dev)
    ou.ops.set_permutation_table(n, kw['perms'].to(dev))
    d = {k: v.to(dev) for k, v in kw.items() if k != 'perms'}
    leaves = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')

    def fwd(seed):
        g = dict(d)
        for k in leaves:
            g[k] = d[k].clone().requires_grad_(True)
        out = ou.optix_env_shade(ctx, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                                 g['light'], g['pdf'], g['rows'], g['cols'], BSDF='pbr', n_samples_x=n, rnd_seed=seed)
        return g, out

    def grads(g, out):
        (out[0].sum() + 2.0 * out[1].sum()).backward()
        return [g[k].grad.clone() for k in leaves]

    g, out = fwd(7)
    ref = grads(g, out)                                   # plain forward -> backward (stream re-used)
    g, out = fwd(7)
    fwd(8)                                                # another launch overwrites the context's stream
    got = grads(g, out)
    for a, b, k in zip(got[:4], ref[:4], leaves):
        assert torch.equal(a, b), k
    assert_close(got[4], ref[4], 1e-4, floor=1e-3 * max(1.0, ref[4].abs().max().item()))
    g, out = fwd(7)
    ou.optix_build_bvh(ctx, m['v_pos'].to(dev), m['t_pos_idx'].to(dev), rebuild=1)   # same geometry, new build
    got = grads(g, out)
    for a, b, k in zip(got[:4], ref[:4], leaves):
        assert torch.equal(a, b), k
    # decorrelated seeds (rnd_seed=None, ops.py:83,99): forward and backward draw independent samples; must run and be finite
    g, out = fwd(None)
    got = grads(g, out)
    assert all(torch.isfinite(x).all() for x in got) and got[1].abs().sum().item() > 0 and got[4].abs().sum().item() > 0


def test_decorrelated_seeds_forward_and_backward_vs_oracle(dev):
    """rnd_seed=None (FLAGS.decorrelated; render/optixutils/ops.py:83,99): the forward pass and the backward pass each draw a fresh seed from
    numpy's global generator, in that order.  With that generator seeded, the forward image must be the oracle's at the first draw and the
    gradients the oracle's at the SECOND (other samples, other visibility: nothing of the forward's stream may be reused)."""
    from nvdiffrecmc_amd import optixutils as ou
    H = W = 48
    n = 4
    inp = scene_cpu.make_inputs('bob', H, W, n, view=5, probe_res=64, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    m = inp['mesh']
    ctx = make_ctx(m, dev)
    ou.ops.set_permutation_table(n, kw['perms'].to(dev))
    g = torch.Generator().manual_seed(21)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    np.random.seed(20240607)
    s_fwd, s_bwd = int(np.random.randint(2**31)), int(np.random.randint(2**31))
    assert s_fwd != s_bwd
    for cache_vis in (False, True):         # (the visibility cache must not be consulted either: the seeds differ)
        ou.ops._optix_env_shade_func.cache_visibility = cache_vis
        d = {k: v.to(dev) for k, v in kw.items() if k != 'perms'}
        leaves = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')
        for k in leaves:
            d[k] = d[k].clone().requires_grad_(True)
        np.random.seed(20240607)
        diff, spec = ou.optix_env_shade(ctx, d['mask'], d['ro'], d['gb_pos'], d['gb_normal'], d['gb_view_pos'], d['gb_kd'], d['gb_ks'],
                                        d['light'], d['pdf'], d['rows'], d['cols'], BSDF='pbr', n_samples_x=n, rnd_seed=None)
        ((diff * dg.to(dev)).sum() + (spec * sg.to(dev)).sum()).backward()
        ref_f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=s_fwd, n_threads=NT)
        ref_b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=s_bwd, diff_grad=dg, spec_grad=sg, n_threads=NT)
        assert_close(diff, ref_f['diff'], FWD_RTOL, what='diff')
        assert_close(spec, ref_f['spec'], FWD_RTOL, what='spec')
        for k in leaves:
            ref = ref_b[k + '_grad']
            assert_close(d[k].grad, ref, BWD_RTOL, floor=1e-3 * max(1.0, ref.abs().max().item()), what=k)
        # and they are NOT the gradients of the forward's samples
        same = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=s_fwd, diff_grad=dg, spec_grad=sg, n_threads=NT)
        assert (same['gb_normal_grad'] - ref_b['gb_normal_grad']).abs().max().item() > 1e-3 * ref_b['gb_normal_grad'].abs().max().item()
    ou.ops._optix_env_shade_func.cache_visibility = True


def test_stage_profiling_hooks(dev):
    """nvdr_ctx_set_profiling / nvdr_env_shade_stage_times (the HIP-event timing bench.py's roofline uses)."""
    H = W = 64
    n = 4
    inp = scene_cpu.make_inputs('bob', H, W, n, view=1, probe_res=64, n_threads=NT)
    kw = scene_cpu.shade_kwargs(inp)
    ctx = make_ctx(inp['mesh'], dev)
    g = torch.Generator().manual_seed(1)
    dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    ctx.set_profiling(True)
    for seed in range(3):
        gpu_env_shade(ctx, kw, dev, 'pbr', n, seed, dg, sg)
    nf, f = ctx.stage_times(backward=False)
    nb, b = ctx.stage_times(backward=True)
    ctx.set_profiling(False)
    assert nf == 3 and nb == 3
    assert all(t > 0 for t in f) and b[1] > 0 and b[2] > 0
    assert b[0] < f[0]          # backward re-used the stream: no sample generation
    nf, f = ctx.stage_times(backward=False)
    assert nf == 0
