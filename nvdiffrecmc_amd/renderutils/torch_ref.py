"""Plain-torch formulations of the renderutils operators: the `use_python=True` switch of the
reference API (render/renderutils/ops.py, backed there by bsdf.py:19-151 and loss.py:15-47).

These run wherever their tensors live and are differentiated by autograd.  They are an explicit,
opt-in validation path of the public API -- the default path (use_python=False) is always the HIP
kernel and never falls back to this module.
"""
import math

import torch

_EPS = 1e-4            # specular_epsilon
_BEND_THRESHOLD = 0.1  # NORMAL_THRESHOLD


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def _unit(v):
    return torch.nn.functional.normalize(v, dim=-1)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl):
    n, t = _unit(smooth_nrm), _unit(smooth_tng)
    view = _unit(view_pos - pos)
    b = _unit(torch.cross(t, n, dim=-1))
    sgn = -1.0 if opengl else 1.0
    shading = _unit(t * perturbed_nrm[..., 0:1] + sgn * b * perturbed_nrm[..., 1:2]
                    + n * perturbed_nrm[..., 2:3].clamp(min=0.0))
    if two_sided_shading:
        front = _dot(geom_nrm, view) > 0
        shading = torch.where(front, shading, -shading)
        geom_nrm = torch.where(front, geom_nrm, -geom_nrm)
    w = (_dot(view, shading) / _BEND_THRESHOLD).clamp(0, 1)
    return torch.lerp(geom_nrm, shading, w)


def lambert(nrm, wi):
    return _dot(nrm, wi).clamp(min=0.0) / math.pi


def fresnel_shlick(f0, f90, cos_theta):
    c = cos_theta.clamp(_EPS, 1.0 - _EPS)
    return f0 + (f90 - f0) * (1.0 - c) ** 5.0


def frostbite(nrm, wi, wo, linear_roughness):
    wi_n, wo_n = _dot(wi, nrm), _dot(wo, nrm)
    wi_h = _dot(wi, _unit(wo + wi))
    f90 = 0.5 * linear_roughness + 2.0 * wi_h * wi_h * linear_roughness
    res = fresnel_shlick(1.0, f90, wi_n) * fresnel_shlick(1.0, f90, wo_n) * (1.0 - (0.51 / 1.51) * linear_roughness)
    return torch.where((wi_n > 0.0) & (wo_n > 0.0), res, torch.zeros_like(res))


def ndf_ggx(alpha_sqr, cos_theta):
    c = cos_theta.clamp(_EPS, 1.0 - _EPS)
    d = (c * alpha_sqr - c) * c + 1
    return alpha_sqr / (d * d * math.pi)


def lambda_ggx(alpha_sqr, cos_theta):
    c2 = cos_theta.clamp(_EPS, 1.0 - _EPS) ** 2
    return 0.5 * (torch.sqrt(1 + alpha_sqr * (1.0 - c2) / c2) - 1.0)


def masking_smith(alpha_sqr, cos_i, cos_o):
    return 1 / (1 + lambda_ggx(alpha_sqr, cos_i) + lambda_ggx(alpha_sqr, cos_o))


def pbr_specular(col, nrm, wo, wi, alpha, min_roughness=0.08):
    a2 = alpha.clamp(min_roughness * min_roughness, 1.0) ** 2
    h = _unit(wo + wi)
    wo_n, wi_n = _dot(wo, nrm), _dot(wi, nrm)
    w = fresnel_shlick(col, 1, _dot(wo, h)) * ndf_ggx(a2, _dot(nrm, h)) * masking_smith(a2, wo_n, wi_n) * 0.25 \
        / wo_n.clamp(min=_EPS)
    return torch.where((wo_n > _EPS) & (wi_n > _EPS), w, torch.zeros_like(w))


def pbr_bsdf(kd, arm, pos, nrm, view_pos, light_pos, min_roughness, BSDF):
    wo, wi = _unit(view_pos - pos), _unit(light_pos - pos)
    occ, rough, metal = arm[..., 0:1], arm[..., 1:2], arm[..., 2:3]
    spec_col = (0.04 * (1.0 - metal) + kd * metal) * (1 - occ)
    diff_col = kd * (1.0 - metal)
    diffuse = diff_col * (lambert(nrm, wi) if BSDF == 0 else frostbite(nrm, wi, wo, rough))
    return diffuse + pbr_specular(spec_col, nrm, wo, wi, rough * rough, min_roughness=min_roughness)


def _tonemap_log_srgb(x):
    f = torch.log(x.clamp(0, 65535) + 1)
    return torch.where(f > 0.0031308, f.clamp(min=0.0031308) ** (1.0 / 2.4) * 1.055 - 0.055, 12.92 * f)


def image_loss(img, target, loss, tonemapper):
    if tonemapper == 'log_srgb':
        img, target = _tonemap_log_srgb(img), _tonemap_log_srgb(target)
    d = img - target
    if loss == 'mse':
        return (d * d).mean()
    if loss == 'smape':
        return (d.abs() / (img.abs() + target.abs() + 0.01)).mean()
    if loss == 'relmse':
        return (d * d / (img * img + target * target + 0.01)).mean()
    if loss == 'n2n':
        return (d * d / (img.detach() ** 2 + 0.01)).mean()
    return d.abs().mean()


def shade_composite(diff, spec, kd, ks, bsdf='pbr'):
    """render/render.py:119-127 with the division of optixutils/ops.py:139-141 in front."""
    d = diff[..., 0:3] / diff[..., 3:4] if diff.shape[-1] == 4 else diff
    s = spec[..., 0:3] / spec[..., 3:4] if spec.shape[-1] == 4 else spec
    if bsdf == 'pbr':
        return d * (kd * (1.0 - ks[..., 2:3])) + s
    return d * kd
