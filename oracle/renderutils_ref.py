"""CPU ORACLE for the renderutils operators (test infrastructure, NOT product code).

A torch restatement (CPU tensors, autograd for the gradients) of the reference's pure-PyTorch
implementations, each function citing what it follows:
    render/renderutils/bsdf.py:19-151   (shading normal, lambert, frostbite, fresnel, ndf, lambda, masking,
                                         pbr_specular, pbr_bsdf)
    render/renderutils/loss.py:15-47    (tonemapper + the five image losses)
    render/renderutils/ops.py:532,549   (xfm_points / xfm_vectors as matmul)
Pinned in this container against the reference's own module imported from /root/reference
(tests/test_oracle_pins.py) and through the committed vectors tests/golden/renderutils_*.npz, which
tools/make_golden.py generates by running that module.
"""
import math

import torch

NORMAL_THRESHOLD = 0.1   # bsdf.py:13
specular_epsilon = 1e-4  # bsdf.py:94


def dot(x, y):                      # bsdf.py:19
    return torch.sum(x * y, -1, keepdim=True)


def safe_normalize(x):              # bsdf.py:25 (torch.nn.functional.normalize, eps 1e-12)
    return x / torch.clamp(torch.sqrt(dot(x, x)), min=1e-12)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True):
    """bsdf.py:28-51 (_bend_normal, _perturb_normal, bsdf_prepare_shading_normal)"""
    smooth_nrm = safe_normalize(smooth_nrm)
    smooth_tng = safe_normalize(smooth_tng)
    view_vec = safe_normalize(view_pos - pos)
    bitang = safe_normalize(torch.cross(smooth_tng, smooth_nrm, dim=-1))
    sign = -1.0 if opengl else 1.0
    shading = smooth_tng * perturbed_nrm[..., 0:1] + sign * bitang * perturbed_nrm[..., 1:2] \
        + smooth_nrm * torch.clamp(perturbed_nrm[..., 2:3], min=0.0)
    shading = safe_normalize(shading)
    if two_sided_shading:
        facing = dot(geom_nrm, view_vec) > 0
        shading = torch.where(facing, shading, -shading)
        geom_nrm = torch.where(facing, geom_nrm, -geom_nrm)
    t = torch.clamp(dot(view_vec, shading) / NORMAL_THRESHOLD, min=0, max=1)
    return torch.lerp(geom_nrm, shading, t)


def lambert(nrm, wi):               # bsdf.py:57
    return torch.clamp(dot(nrm, wi), min=0.0) / math.pi


def fresnel_shlick(f0, f90, cosTheta):      # bsdf.py:96
    c = torch.clamp(cosTheta, min=specular_epsilon, max=1.0 - specular_epsilon)
    return f0 + (f90 - f0) * (1.0 - c) ** 5.0


def frostbite(nrm, wi, wo, linearRoughness):    # bsdf.py:64
    wiDotN, woDotN = dot(wi, nrm), dot(wo, nrm)
    h = safe_normalize(wo + wi)
    wiDotH = dot(wi, h)
    energyBias = 0.5 * linearRoughness
    energyFactor = 1.0 - (0.51 / 1.51) * linearRoughness
    f90 = energyBias + 2.0 * wiDotH * wiDotH * linearRoughness
    res = fresnel_shlick(1.0, f90, wiDotN) * fresnel_shlick(1.0, f90, woDotN) * energyFactor
    return torch.where((wiDotN > 0.0) & (woDotN > 0.0), res, torch.zeros_like(res))


def ndf_ggx(alphaSqr, cosTheta):    # bsdf.py:100
    c = torch.clamp(cosTheta, min=specular_epsilon, max=1.0 - specular_epsilon)
    d = (c * alphaSqr - c) * c + 1
    return alphaSqr / (d * d * math.pi)


def lambda_ggx(alphaSqr, cosTheta):  # bsdf.py:105
    c = torch.clamp(cosTheta, min=specular_epsilon, max=1.0 - specular_epsilon)
    c2 = c * c
    tan2 = (1.0 - c2) / c2
    return 0.5 * (torch.sqrt(1 + alphaSqr * tan2) - 1.0)


def masking_smith(alphaSqr, cosThetaI, cosThetaO):  # bsdf.py:112
    return 1 / (1 + lambda_ggx(alphaSqr, cosThetaI) + lambda_ggx(alphaSqr, cosThetaO))


def pbr_specular(col, nrm, wo, wi, alpha, min_roughness=0.08):  # bsdf.py:117
    _alpha = torch.clamp(alpha, min=min_roughness * min_roughness, max=1.0)
    alphaSqr = _alpha * _alpha
    h = safe_normalize(wo + wi)
    woDotN, wiDotN, woDotH, nDotH = dot(wo, nrm), dot(wi, nrm), dot(wo, h), dot(nrm, h)
    D = ndf_ggx(alphaSqr, nDotH)
    G = masking_smith(alphaSqr, woDotN, wiDotN)
    F = fresnel_shlick(col, 1, woDotH)
    w = F * D * G * 0.25 / torch.clamp(woDotN, min=specular_epsilon)
    return torch.where((woDotN > specular_epsilon) & (wiDotN > specular_epsilon), w, torch.zeros_like(w))


def pbr_bsdf(kd, arm, pos, nrm, view_pos, light_pos, min_roughness=0.08, bsdf='lambert'):  # bsdf.py:137
    wo = safe_normalize(view_pos - pos)
    wi = safe_normalize(light_pos - pos)
    spec_str, roughness, metallic = arm[..., 0:1], arm[..., 1:2], arm[..., 2:3]
    ks = (0.04 * (1.0 - metallic) + kd * metallic) * (1 - spec_str)
    kd = kd * (1.0 - metallic)
    diffuse = kd * (lambert(nrm, wi) if bsdf == 'lambert' else frostbite(nrm, wi, wo, roughness))
    return diffuse + pbr_specular(ks, nrm, wo, wi, roughness * roughness, min_roughness=min_roughness)


def tonemap_srgb(f):                # loss.py:15
    return torch.where(f > 0.0031308, torch.pow(torch.clamp(f, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * f)


def image_loss(img, target, loss='l1', tonemapper='none', kernel_semantics=False):      # loss.py:33-47
    """kernel_semantics=True adds what the reference's CUDA kernel does and its python path does not: BOTH images are
    clamped to [0, 65535] for every tonemapper (render/renderutils/c_src/loss.cu:113-114), so out-of-range HDR values
    contribute a clamped value and get zero gradient (loss.cu:221-226).  In range the two paths agree."""
    if kernel_semantics:
        img = torch.clamp(img, min=0, max=65535)
        target = torch.clamp(target, min=0, max=65535)
    if tonemapper == 'log_srgb':
        img = tonemap_srgb(torch.log(torch.clamp(img, min=0, max=65535) + 1))
        target = tonemap_srgb(torch.log(torch.clamp(target, min=0, max=65535) + 1))
    if loss == 'mse':
        return torch.nn.functional.mse_loss(img, target)
    if loss == 'smape':         # loss.py:18
        return torch.mean(torch.abs(img - target) / (torch.abs(img) + torch.abs(target) + 0.01))
    if loss == 'relmse':        # loss.py:23
        return torch.mean((img - target) ** 2 / (img * img + target * target + 0.01))
    if loss == 'n2n':           # loss.py:28
        return torch.mean((img - target) ** 2 / (img.detach() * img.detach() + 0.01))
    return torch.nn.functional.l1_loss(img, target)


def xfm_points(points, matrix):     # ops.py:532
    return torch.matmul(torch.nn.functional.pad(points, pad=(0, 1), mode='constant', value=1.0), torch.transpose(matrix, 1, 2))


def xfm_vectors(vectors, matrix):   # ops.py:549
    return torch.matmul(torch.nn.functional.pad(vectors, pad=(0, 1), mode='constant', value=0.0),
                        torch.transpose(matrix, 1, 2))[..., 0:3].contiguous()


def bilateral_denoiser_torch(col, nrm, zdz, sigma):
    """The authors' own torch formulation of the denoiser (render/optixutils/tests/filter_test.py:31-74),
    restated with explicit zero padding instead of torch.roll + mask; differentiable w.r.t. col."""
    eps = 0.0001
    R = 2 * math.ceil(sigma * 2.5) + 1
    N, H, W, _ = col.shape
    pad = lambda t: torch.nn.functional.pad(t.permute(0, 3, 1, 2), (R, R, R, R)).permute(0, 2, 3, 1)
    colp, nrmp, zdzp = pad(col), pad(nrm), pad(zdz)
    valid = pad(torch.ones(N, H, W, 1))
    acc = torch.zeros_like(col)
    accw = torch.zeros(N, H, W, 1)
    for fy in range(-R, R + 1):
        for fx in range(-R, R + 1):
            sl = (slice(None), slice(R + fy, R + fy + H), slice(R + fx, R + fx + W))
            d2 = float(fx * fx + fy * fy)
            dist = math.sqrt(d2)
            w_xy = math.exp(-d2 / (2 * sigma * sigma))
            with torch.no_grad():
                w_n = torch.pow(torch.clamp(dot(nrmp[sl], nrm), min=eps, max=1.0), 128.0)
                w_d = torch.exp(-(torch.abs(zdzp[sl][..., 0:1] - zdz[..., 0:1]) / torch.clamp(zdz[..., 1:2] * dist, min=eps)))
                w = w_xy * w_n * w_d * valid[sl]
            acc = acc + colp[sl] * w
            accw = accw + w
    return acc / torch.clamp(accw, min=eps)
