"""Self-consistency of the CPU oracle: properties the raygen program must have WHATEVER it is compared with --
the only protection against a mistake that the restatement and the reference-on-CPU build (which share the shim and the
triangle predicate) would make together.

  * MIS unbiasedness (kernel.cu:409,507-530): the estimator  sum_i f L V / (p_light + p_bsdf)  over one light-sampled and one
    BSDF-sampled direction per stratum is the balance heuristic; it is unbiased only if BOTH pdfs are the true densities of
    BOTH sampling procedures.  Checked three ways on an unoccluded patch: the combined estimator against a brute-force
    quadrature of the integral over the probe's texels, and the two single-technique estimators (light samples weighted
    1/p_light, BSDF samples weighted 1/p_bsdf) against each other.
  * finite differences of the backward pass (kernel.cu:422-457) with FROZEN samples: the reference differentiates the evaluation
    of the samples only (not their directions, pdfs or visibility), so the samples are generated from an unperturbed copy of the
    G-buffer (oracle_env_shade_frozen) while kd / ks / normal / position / radiance move.
No GPU."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc, renderutils_ref as rr
from nvdiffrecmc_amd import scene as sc

NT = min(orc.max_threads(), 16)
FAR_TRI = (torch.tensor([[100.0, -50.0, 0.0], [101.0, -50.0, 0.0], [100.0, -50.0, 1.0]]), torch.tensor([[0, 1, 2]], dtype=torch.int32))


def _patch(H, W, nrm, view, kd, ks, base):
    """H x W pixels that all see the same surface point (every pixel draws its own samples: kernel.cu:504)."""
    pdf, cols, rows = orc.light_update_pdf(base)
    n = torch.nn.functional.normalize(torch.tensor(nrm), dim=0)
    kw = dict(mask=torch.ones(1, H, W), gb_pos=torch.zeros(1, H, W, 3), gb_normal=n.expand(1, H, W, 3).contiguous(),
              gb_view_pos=torch.tensor(view).view(1, 1, 1, 3), gb_kd=torch.tensor(kd).expand(1, H, W, 3).contiguous(),
              gb_ks=torch.tensor(ks).expand(1, H, W, 3).contiguous(), light=base, pdf=pdf, rows=rows, cols=cols)
    kw['ro'] = (kw['gb_pos'] + kw['gb_normal'] * 1e-3).contiguous()
    return kw, n


def _probe_dirs(Hq, Wq):
    """Quadrature nodes of the lat-long parametrisation (kernel.cu:131-138) and their solid angles."""
    v = (torch.arange(Hq, dtype=torch.float64) + 0.5) / Hq
    u = (torch.arange(Wq, dtype=torch.float64) + 0.5) / Wq
    theta, phi = (v * math.pi)[:, None], ((2 * u - 1) * math.pi)[None, :]
    d = torch.stack([torch.sin(theta) * torch.sin(phi), torch.cos(theta).expand(Hq, Wq), -torch.sin(theta) * torch.cos(phi)], -1)
    dw = torch.sin(theta).expand(Hq, Wq) * (math.pi / Hq) * (2 * math.pi / Wq)
    return d, dw, u, v


def _light_pdf(dirs, pdf):
    """lightPDF(dir) of kernel.cu:171-182 in float64 torch ops (nearest texel of the direction's lat-long coordinates)."""
    Hl, Wl = pdf.shape
    u = torch.atan2(dirs[..., 0], -dirs[..., 2]) / (2 * math.pi) + 0.5
    v = torch.acos(torch.clamp(dirs[..., 1], -1, 1)) / math.pi
    x = torch.clamp((u * Wl).long(), 0, Wl - 1)
    y = torch.clamp((v * Hl).long(), 0, Hl - 1)
    return pdf.double()[y, x] * (Hl * Wl) / (2 * math.pi ** 2 * torch.clamp(torch.sin(v * math.pi), min=1e-4))


@pytest.mark.parametrize('name,bsdf,ks', [('white_rough', 'white', (0.0, 0.6, 0.0)), ('white_glossy_sampler', 'white', (0.0, 0.25, 0.7)),
                                          ('pbr_rough', 'pbr', (0.0, 0.5, 0.0)), ('pbr_metal', 'pbr', (0.1, 0.35, 1.0))])
def test_mis_estimator_is_unbiased(name, bsdf, ks):
    H = W = 24
    n_s = 16                                                   # 256 strata x 576 pixels: 147 k light + 147 k BSDF samples
    S = n_s * n_s
    base = sc.env_map('E1', 32)
    kd = (0.7, 0.5, 0.3)
    kw, nrm = _patch(H, W, [0.3, 1.0, 0.2], [0.4, 1.5, 1.2], kd, ks, base)
    out = orc.env_shade(*FAR_TRI, **kw, perms=sc.perms_table(n_s), bsdf=bsdf, n_samples_x=n_s, rnd_seed=7, n_threads=NT, want_dbg=True,
                        want_vis=True)
    assert out['vis'].min().item() == 1                        # nothing occludes the patch
    # ---- the integral itself: quadrature over 8 x 8 nodes per texel (L is piecewise constant per texel: nearest lookup, kernel.cu:195-201)
    Hl, Wl = base.shape[0], base.shape[1]
    d, dw, u, v = _probe_dirs(Hl * 8, Wl * 8)
    L = base.double()[torch.clamp((v * Hl).long(), 0, Hl - 1)[:, None], torch.clamp((u * Wl).long(), 0, Wl - 1)[None, :]]   # [Hq,Wq,3]
    nd = nrm.double()
    lam = torch.clamp((d * nd).sum(-1), min=0.0) / math.pi
    diff_q = (lam[..., None] * L * dw[..., None]).sum((0, 1))
    est = out['diff'].double().mean((0, 1, 2))
    assert torch.allclose(est, diff_q, rtol=0.02), (name, est, diff_q)
    if bsdf == 'pbr':
        wo = torch.nn.functional.normalize(kw['gb_view_pos'].double().view(3) - 0.0, dim=0)
        kdt, kst = torch.tensor(kd, dtype=torch.float64), torch.tensor(ks, dtype=torch.float64)
        spec_col = (0.04 * (1 - kst[2]) + kdt * kst[2]) * (1 - kst[0])                                   # bsdf.h:227
        f = rr.pbr_specular(spec_col.expand(*d.shape[:2], 3), nd.expand_as(d), wo.expand_as(d), d, (kst[1] ** 2).expand(*d.shape[:2], 1))
        spec_q = (f * L * dw[..., None]).sum((0, 1))
        est_s = out['spec'].double().mean((0, 1, 2))
        assert torch.allclose(est_s, spec_q, rtol=0.03), (name, est_s, spec_q)
    # ---- single-technique estimators from the SAME samples: light samples / p_light and BSDF samples / p_bsdf (p_bsdf = pdf sum - p_light)
    dbg = out['dbg'].double().view(H * W, S, 2, 4)
    dirs, psum = dbg[..., 0:3], dbg[..., 3]
    pl = _light_pdf(dirs, kw['pdf'])
    pb = psum - pl
    Ls = base.double()
    uu = torch.atan2(dirs[..., 0], -dirs[..., 2]) / (2 * math.pi) + 0.5
    vv = torch.acos(torch.clamp(dirs[..., 1], -1, 1)) / math.pi
    Lsmp = Ls[torch.clamp((vv * Hl).long(), 0, Hl - 1), torch.clamp((uu * Wl).long(), 0, Wl - 1)]        # [P,S,2,3]
    lam_s = torch.clamp((dirs * nd).sum(-1), min=0.0) / math.pi
    fL = lam_s[..., None] * Lsmp
    ok_l, ok_b = pl[:, :, 0] > 1e-6, pb[:, :, 1] > 1e-6
    e_light = (fL[:, :, 0] / pl[:, :, 0, None].clamp(min=1e-6) * ok_l[..., None]).mean((0, 1))
    e_bsdf = (fL[:, :, 1] / pb[:, :, 1, None].clamp(min=1e-6) * ok_b[..., None]).mean((0, 1))
    e_mis = (fL / psum[..., None].clamp(min=1e-4)).sum(2).mean((0, 1))
    assert torch.allclose(e_mis, est, rtol=2e-3)               # the python re-evaluation of the recorded samples IS the estimator
    assert torch.allclose(e_light, diff_q, rtol=0.03), (name, 'light-only', e_light, diff_q)
    assert torch.allclose(e_bsdf, diff_q, rtol=0.03), (name, 'bsdf-only', e_bsdf, diff_q)
    assert torch.allclose(e_light, e_bsdf, rtol=0.04), (name, e_light, e_bsdf)


def _fd_scene(H=12, W=12, n_s=4, seed=3):
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.rand(*s, generator=g)
    base = sc.env_map('E1', 32)
    pdf, cols, rows = orc.light_update_pdf(base)
    nrm = torch.nn.functional.normalize(torch.tensor([0.1, 1.0, 0.1]) + 0.5 * (R(1, H, W, 3) - 0.5), dim=-1).contiguous()
    kw = dict(mask=torch.ones(1, H, W), gb_pos=(0.2 * (R(1, H, W, 3) - 0.5)).contiguous(), gb_normal=nrm,
              gb_view_pos=torch.tensor([0.3, 1.2, 0.9]).view(1, 1, 1, 3), gb_kd=(0.2 + 0.6 * R(1, H, W, 3)).contiguous(),
              gb_ks=torch.stack([0.2 * R(1, H, W), 0.35 + 0.4 * R(1, H, W), 0.8 * R(1, H, W)], -1).contiguous(),
              light=base, pdf=pdf, rows=rows, cols=cols)
    kw['ro'] = (kw['gb_pos'] + kw['gb_normal'] * 1e-3).contiguous()
    dg, sg = R(1, H, W, 3), R(1, H, W, 3)
    return kw, dg, sg, sc.perms_table(n_s), n_s


@pytest.mark.parametrize('bsdf', ['pbr', 'diffuse'])
def test_backward_matches_finite_differences_with_frozen_samples(bsdf):
    kw, dg, sg, perms, n_s = _fd_scene()
    frozen = {k: kw[k].clone() for k in ('gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks')}
    common = dict(perms=perms, bsdf=bsdf, n_samples_x=n_s, rnd_seed=11, n_threads=NT, frozen=frozen)

    def loss(**over):
        o = orc.env_shade(*FAR_TRI, **dict(kw, **over), **common)
        return float((o['diff'].double() * dg.double()).sum() + (o['spec'].double() * sg.double()).sum())

    b = orc.env_shade(*FAR_TRI, **kw, **dict(common, n_threads=1), diff_grad=dg, spec_grad=sg)     # one thread: a fixed order of the light-gradient additions
    # the frozen-sample program at the unperturbed point IS the reference program
    b0 = orc.env_shade(*FAR_TRI, **kw, perms=perms, bsdf=bsdf, n_samples_x=n_s, rnd_seed=11, n_threads=1, diff_grad=dg, spec_grad=sg)
    for k in b0:
        if k != 'covered':
            assert torch.equal(b[k], b0[k]), k
    g = torch.Generator().manual_seed(5)
    checked = 0
    for key, gkey, h in (('gb_kd', 'gb_kd_grad', 2e-3), ('gb_ks', 'gb_ks_grad', 1e-3), ('gb_normal', 'gb_normal_grad', 1e-3),
                         ('gb_pos', 'gb_pos_grad', 1e-3), ('light', 'light_grad', None)):
        grad = b[gkey].double()
        if bsdf == 'diffuse' and key in ('gb_kd', 'gb_ks', 'gb_pos'):
            assert float(grad.abs().max()) == 0.0              # the Lambert-only evaluation does not depend on them (kernel.cu:440-444)
            continue
        for trial in range(3):
            delta = torch.rand(kw[key].shape, generator=g) - 0.5
            if key == 'light':
                # move the texels that carry gradient, relative to their radiance (the suns are 1e3 x the sky)
                delta = delta * kw['light'] * (grad.abs().sum(-1, keepdim=True) > 0)
                hh = 1e-2
            else:
                hh = h
            up = (kw[key] + hh * delta).contiguous()
            dn = (kw[key] - hh * delta).contiguous()
            fd = (loss(**{key: up}) - loss(**{key: dn})) / (2 * hh)
            an = float((grad * delta.double()).sum())
            scale = float((grad.abs() * delta.double().abs()).sum()) + 1e-12
            assert abs(fd - an) <= 0.02 * scale, '%s %s trial %d: finite difference %.6g vs analytic %.6g (scale %.3g)' % (bsdf, key, trial, fd, an, scale)
            assert abs(an) > 1e-4 * scale or key == 'gb_pos'
            checked += 1
    assert checked >= (15 if bsdf == 'pbr' else 6)


def test_light_gradient_is_exactly_linear():
    """The radiance enters every output linearly (kernel.cu:203-211,409): with frozen tables the light gradient is the
    EXACT derivative -- the loss at light + t * delta is affine in t."""
    kw, dg, sg, perms, n_s = _fd_scene(seed=9)
    common = dict(perms=perms, bsdf='pbr', n_samples_x=n_s, rnd_seed=2, n_threads=NT)
    b = orc.env_shade(*FAR_TRI, **kw, **common, diff_grad=dg, spec_grad=sg)
    g = torch.Generator().manual_seed(1)
    delta = (torch.rand(kw['light'].shape, generator=g) * kw['light']).contiguous()

    def loss(t):
        o = orc.env_shade(*FAR_TRI, **dict(kw, light=(kw['light'] + t * delta).contiguous()), **common)
        return float((o['diff'].double() * dg.double()).sum() + (o['spec'].double() * sg.double()).sum())
    l0, l1, l2 = loss(0.0), loss(0.5), loss(1.0)
    an = float((b['light_grad'].double() * delta.double()).sum())
    assert abs((l2 - l0) - an) < 2e-4 * abs(an) and abs((l1 - l0) - 0.5 * an) < 2e-4 * abs(an)
