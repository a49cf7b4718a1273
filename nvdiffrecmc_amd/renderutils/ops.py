"""Drop-in replacement for render/renderutils/ops.py on MI355X.

Same twelve public names, argument order, defaults, `use_python=` switch, broadcasting rules and
anomaly-mode finite checks as the reference module (ops.py:89-551); the CUDA plugin
(renderutils_plugin) is replaced by libnvdr_hip.so (csrc/renderutils.hip).  Broadcast inputs
get full-extent gradients from the kernels; autograd sums them back to the input shape, exactly
as with the reference plugin (c_src/tensor.h:60-62).

Deliberate deviations: loss='n2n' really computes N2N (the CUDA path of the
reference silently turns it into L1, torch_bindings.cpp:727-737); perturbed_nrm=None creates its
default on the device of `pos` instead of the hard-coded 'cuda' (ops.py:218).
"""
import ctypes

import torch

from .. import _lib
from . import torch_ref

__all__ = ["xfm_vectors", "xfm_points", "image_loss", "prepare_shading_normal", "lambert", "frostbite_diffuse",
           "pbr_specular", "pbr_bsdf", "_fresnel_shlick", "_ndf_ggx", "_lambda_ggx", "_masking_smith"]


def _extent(*ts):
    return tuple(max(t.shape[d] for t in ts) for d in range(3))


def _check4(t, name, channels):
    _lib.require_cuda_f32(t, name)
    if t.dim() != 4 or t.shape[3] not in (channels, 1):
        # same condition as CHECK_TENSOR in torch_bindings.cpp:24-28
        raise RuntimeError("%s must have shape [>0, >0, >0, %d] (got %s)" % (name, channels, tuple(t.shape)))


def _views(*ts):
    vs = [_lib.tensor_view(t) for t in ts]
    return vs, [ctypes.byref(v) for v in vs]


def _finite(out, name):
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of %s contains inf or NaN" % name
    return out


def _make_op(fn_name, in_channels, out_channels, extra_fwd=0):
    """autograd.Function for an element-wise op `nvdr_<fn_name>_{fwd,bwd}` with tensor inputs of the given
    channel counts, `extra_fwd` trailing non-tensor arguments, one output."""
    n_in = len(in_channels)

    class _Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *args):
            tensors, extra = args[:n_in], args[n_in:]
            for i, (t, c) in enumerate(zip(tensors, in_channels)):
                _check4(t, '%s arg %d' % (fn_name, i), c)
            lib = _lib.load()
            N, H, W = _extent(*tensors)
            out = torch.empty(N, H, W, out_channels, dtype=torch.float32, device=tensors[0].device)
            keep, refs = _views(*tensors)
            rc = getattr(lib, 'nvdr_%s_fwd' % fn_name)(*refs, *extra, _lib.ptr(out), _lib.stream_ptr())
            _lib.check(rc, fn_name + '_fwd')
            ctx.save_for_backward(*tensors)
            ctx.extra = extra
            return out

        @staticmethod
        def backward(ctx, dout):
            tensors = ctx.saved_tensors
            lib = _lib.load()
            N, H, W = _extent(*tensors)
            dout = dout.contiguous()
            grads = [torch.empty(N, H, W, c, dtype=torch.float32, device=dout.device) for c in in_channels]
            keep, refs = _views(*tensors, dout)
            rc = getattr(lib, 'nvdr_%s_bwd' % fn_name)(*refs[:-1], *ctx.extra, refs[-1], *[_lib.ptr(g) for g in grads],
                                                       _lib.stream_ptr())
            _lib.check(rc, fn_name + '_bwd')
            # a 1-channel input read through a 3-channel slot gets a 3-channel gradient: fold it
            grads = [g.sum(-1, keepdim=True) if t.shape[3] == 1 and g.shape[3] != 1 else g for g, t in zip(grads, tensors)]
            return tuple(grads) + (None,) * len(ctx.extra)

    _Op.__name__ = '_%s_func' % fn_name
    return _Op


_fresnel_shlick_func = _make_op('fresnel_shlick', (3, 3, 1), 3)
_ndf_ggx_func = _make_op('ndf_ggx', (1, 1), 1)
_lambda_ggx_func = _make_op('lambda_ggx', (1, 1), 1)
_masking_smith_func = _make_op('masking_smith', (1, 1, 1), 1)
_lambert_func = _make_op('lambert', (3, 3), 1)
_frostbite_diffuse_func = _make_op('frostbite', (3, 3, 3, 1), 1)
_pbr_specular_func = _make_op('pbr_specular', (3, 3, 3, 3, 1), 3, extra_fwd=1)
_pbr_bsdf_func = _make_op('pbr_bsdf', (3, 3, 3, 3, 3, 3), 3, extra_fwd=2)


# ----------------------------------------------------------------------------------------------
# internal kernels, used for testing (ops.py:89-176)

def _fresnel_shlick(f0, f90, cosTheta, use_python=False):
    out = torch_ref.fresnel_shlick(f0, f90, cosTheta) if use_python else _fresnel_shlick_func.apply(f0, f90, cosTheta)
    return _finite(out, '_fresnel_shlick')


def _ndf_ggx(alphaSqr, cosTheta, use_python=False):
    out = torch_ref.ndf_ggx(alphaSqr, cosTheta) if use_python else _ndf_ggx_func.apply(alphaSqr, cosTheta)
    return _finite(out, '_ndf_ggx')


def _lambda_ggx(alphaSqr, cosTheta, use_python=False):
    out = torch_ref.lambda_ggx(alphaSqr, cosTheta) if use_python else _lambda_ggx_func.apply(alphaSqr, cosTheta)
    return _finite(out, '_lambda_ggx')


def _masking_smith(alphaSqr, cosThetaI, cosThetaO, use_python=False):
    out = (torch_ref.masking_smith(alphaSqr, cosThetaI, cosThetaO) if use_python
           else _masking_smith_func.apply(alphaSqr, cosThetaI, cosThetaO))
    return _finite(out, '_masking_smith')


# ----------------------------------------------------------------------------------------------
# shading normal setup (ops.py:181-227)

_UNIT_Z = {}   # device -> [1,1,1,3] tangent-space 'no perturbation' normal


class _prepare_shading_normal_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl):
        ts = (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)
        for i, t in enumerate(ts):
            _check4(t, 'prepare_shading_normal arg %d' % i, 3)
        ctx.two_sided_shading, ctx.opengl = two_sided_shading, opengl
        N, H, W = _extent(*ts)
        out = torch.empty(N, H, W, 3, dtype=torch.float32, device=pos.device)
        keep, refs = _views(*ts)
        _lib.check(_lib.load().nvdr_prepare_shading_normal_fwd(*refs, int(bool(two_sided_shading)), int(bool(opengl)),
                                                               _lib.ptr(out), _lib.stream_ptr()),
                   'prepare_shading_normal_fwd')
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, dout):
        ts = ctx.saved_tensors
        N, H, W = _extent(*ts)
        dout = dout.contiguous()
        grads = [torch.empty(N, H, W, 3, dtype=torch.float32, device=dout.device) for _ in range(6)]
        keep, refs = _views(*ts, dout)
        _lib.check(_lib.load().nvdr_prepare_shading_normal_bwd(*refs, int(bool(ctx.two_sided_shading)), int(bool(ctx.opengl)),
                                                               *[_lib.ptr(g) for g in grads], _lib.stream_ptr()),
                   'prepare_shading_normal_bwd')
        return tuple(grads) + (None, None)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True,
                           opengl=True, use_python=False):
    '''Takes care of all corner cases and produces a final normal used for shading: constructs the tangent
    space, flips towards the viewer for two-sided shading, perturbs by the normal map and bends back-facing
    normals (same contract as the reference, ops.py:196-214).  All tensors [N,H,W,3] or broadcastable.'''
    if perturbed_nrm is None:
        # the reference builds this constant with torch.tensor(..., device='cuda') on every call (ops.py:218): a pageable
        # host-to-device copy that makes the host wait for the stream to drain -- 1 ms per iteration with the GPU kept
        # busy, and it stops the host from running ahead.  One constant per device instead.
        perturbed_nrm = _UNIT_Z.get(pos.device)
        if perturbed_nrm is None:
            perturbed_nrm = _UNIT_Z[pos.device] = torch.tensor([0, 0, 1], dtype=torch.float32, device=pos.device,
                                                               requires_grad=False)[None, None, None, ...]
    if use_python:
        out = torch_ref.prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm,
                                               two_sided_shading, opengl)
    else:
        out = _prepare_shading_normal_func.apply(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm,
                                                 two_sided_shading, opengl)
    return _finite(out, 'prepare_shading_normal')


class _shading_frame_func(torch.autograd.Function):
    """(shading normal, unit copy, shadow-ray origin) in one launch; the adjoint is prepare_shading_normal's (normal.cu:136-179).
    The unit copy (the filter's guide: bilateral_denoiser differentiates w.r.t. its colour input only, optixutils/ops.py:119) and
    the ray origin (optix_env_shade returns no gradient for `ro`, ops.py:105) are declared non-differentiable, as in the reference's
    graph, so gradients reach the six inputs through the shading normal alone."""

    @staticmethod
    def forward(ctx, pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl, ro_eps):
        ctx.set_materialize_grads(False)        # no zero tensors for the two outputs that carry no gradient
        ins = (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)
        for name, t in zip(('pos', 'view_pos', 'perturbed_nrm', 'smooth_nrm', 'smooth_tng', 'geom_nrm'), ins):
            _check4(t, 'shading_frame ' + name, 3)
        N, H, W = _extent(*ins)
        out = torch.empty(3, N, H, W, 3, dtype=torch.float32, device=pos.device)
        keep, refs = _views(*ins)
        _lib.check(_lib.load().nvdr_shading_frame_fwd(*refs, int(bool(two_sided_shading)), int(bool(opengl)), float(ro_eps), _lib.ptr(out[0]),
                                                      _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.stream_ptr()), 'shading_frame_fwd')
        ctx.save_for_backward(*ins)
        ctx.two_sided_shading, ctx.opengl = two_sided_shading, opengl
        nrm, unit, ro = out[0], out[1], out[2]
        ctx.mark_non_differentiable(unit, ro)
        return nrm, unit, ro

    @staticmethod
    def backward(ctx, dnrm, _dunit, _dro):
        ts = ctx.saved_tensors
        N, H, W = _extent(*ts)
        if dnrm is None:
            return (None,) * 9
        dnrm = dnrm.contiguous()
        grads = torch.empty(6, N, H, W, 3, dtype=torch.float32, device=dnrm.device)
        keep, refs = _views(*ts, dnrm)
        _lib.check(_lib.load().nvdr_prepare_shading_normal_bwd(*refs, int(bool(ctx.two_sided_shading)), int(bool(ctx.opengl)),
                                                               *[_lib.ptr(grads[k]) for k in range(6)], _lib.stream_ptr()), 'shading_frame_bwd')
        out = []
        for k, t in enumerate(ts):              # broadcast inputs get their gradient at the full extent: fold it (tensor.h:60-62)
            g = grads[k] if ctx.needs_input_grad[k] else None
            if g is not None:
                for d in range(4):
                    if t.shape[d] == 1 and g.shape[d] != 1:
                        g = g.sum(d, keepdim=True)
            out.append(g)
        return tuple(out) + (None, None, None)


def shading_frame(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True, ro_eps=0.001):
    """(shading normal, its unit copy, shadow-ray origin pos + normal * ro_eps) in one launch -- prepare_shading_normal, the
    safe_normalize of the denoiser's guide and the offset of render.py:107.  Additive.  Differentiable through the shading normal
    (one launch backward: prepare_shading_normal_bwd); the unit copy and the ray origin carry no gradient (see _shading_frame_func)."""
    if perturbed_nrm is None:
        perturbed_nrm = _UNIT_Z.get(pos.device)
        if perturbed_nrm is None:
            perturbed_nrm = _UNIT_Z[pos.device] = torch.tensor([0, 0, 1], dtype=torch.float32, device=pos.device,
                                                               requires_grad=False)[None, None, None, ...]
    return _shading_frame_func.apply(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl, ro_eps)


# ----------------------------------------------------------------------------------------------
# BSDF functions (ops.py:232-386)

def lambert(nrm, wi, use_python=False):
    '''Lambertian bsdf -> [N,H,W,1].'''
    out = torch_ref.lambert(nrm, wi) if use_python else _lambert_func.apply(nrm, wi)
    return _finite(out, 'lambert')


def frostbite_diffuse(nrm, wi, wo, linearRoughness, use_python=False):
    '''Frostbite normalized Disney diffuse -> [N,H,W,1].'''
    out = (torch_ref.frostbite(nrm, wi, wo, linearRoughness) if use_python
           else _frostbite_diffuse_func.apply(nrm, wi, wo, linearRoughness))
    return _finite(out, 'lambert')


def pbr_specular(col, nrm, wo, wi, alpha, min_roughness=0.08, use_python=False):
    '''GGX specular lobe -> [N,H,W,3]; alpha is [N,H,W,1].'''
    out = (torch_ref.pbr_specular(col, nrm, wo, wi, alpha, min_roughness=min_roughness) if use_python
           else _pbr_specular_func.apply(col, nrm, wo, wi, alpha, float(min_roughness)))
    return _finite(out, 'pbr_specular')


def pbr_bsdf(kd, arm, pos, nrm, view_pos, light_pos, min_roughness=0.08, bsdf="lambert", use_python=False):
    '''Diffuse ('lambert' | 'frostbite') + GGX specular, point light at light_pos -> [N,H,W,3].'''
    BSDF = 1 if bsdf == 'frostbite' else 0
    out = (torch_ref.pbr_bsdf(kd, arm, pos, nrm, view_pos, light_pos, min_roughness, BSDF) if use_python
           else _pbr_bsdf_func.apply(kd, arm, pos, nrm, view_pos, light_pos, float(min_roughness), BSDF))
    return _finite(out, 'pbr_bsdf')


# ----------------------------------------------------------------------------------------------
# fast image loss (ops.py:463-498)

_LOSS = {'l1': 0, 'mse': 1, 'relmse': 2, 'smape': 3, 'n2n': 4}


class _image_loss_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, target, loss, tonemapper):
        _check4(img, 'img', 3)
        _check4(target, 'target', 3)
        ctx.loss, ctx.tonemapper = loss, tonemapper
        ctx.save_for_backward(img, target)
        lib = _lib.load()
        N, H, W = _extent(img, target)
        n_part = lib.nvdr_image_loss_num_partials(N, H, W)
        out = torch.empty(n_part, dtype=torch.float32, device=img.device)
        keep, refs = _views(img, target)
        _lib.check(lib.nvdr_image_loss_fwd(*refs, _LOSS.get(loss, 0), int(tonemapper == 'log_srgb'), _lib.ptr(out),
                                           _lib.stream_ptr()), 'image_loss_fwd')
        return out

    @staticmethod
    def backward(ctx, dout):
        img, target = ctx.saved_tensors
        lib = _lib.load()
        N, H, W = _extent(img, target)
        dout = dout.contiguous()
        gi = torch.empty(N, H, W, 3, dtype=torch.float32, device=img.device)
        # the target of a training iteration is a constant: its gradient is neither computed nor written then
        gt = torch.empty(N, H, W, 3, dtype=torch.float32, device=img.device) if ctx.needs_input_grad[1] else None
        keep, refs = _views(img, target)
        _lib.check(lib.nvdr_image_loss_bwd(*refs, _LOSS.get(ctx.loss, 0), int(ctx.tonemapper == 'log_srgb'), _lib.ptr(dout),
                                           _lib.ptr(gi), _lib.ptr(gt) if gt is not None else None, _lib.stream_ptr()), 'image_loss_bwd')
        return gi, gt, None, None


class _image_loss_mean_func(torch.autograd.Function):
    """The loss as one scalar in two launches (partials + a fixed-order reduction) and a backward that reads its upstream gradient
    from a device scalar: replaces `torch.sum(partials) / (N H W)` and the five small kernels of its adjoint."""

    @staticmethod
    def forward(ctx, img, target, loss, tonemapper):
        _check4(img, 'img', 3)
        _check4(target, 'target', 3)
        ctx.loss, ctx.tonemapper = loss, tonemapper
        ctx.save_for_backward(img, target)
        lib = _lib.load()
        N, H, W = _extent(img, target)
        part = torch.empty(lib.nvdr_image_loss_num_partials(N, H, W) + 1, dtype=torch.float32, device=img.device)
        keep, refs = _views(img, target)
        _lib.check(lib.nvdr_image_loss_mean_fwd(*refs, _LOSS.get(loss, 0), int(tonemapper == 'log_srgb'), _lib.ptr(part[1:]), _lib.ptr(part),
                                                _lib.stream_ptr()), 'image_loss_mean_fwd')
        return part[0]

    @staticmethod
    def backward(ctx, dout):
        img, target = ctx.saved_tensors
        N, H, W = _extent(img, target)
        dout = dout.contiguous().view(1)
        gi = torch.empty(N, H, W, 3, dtype=torch.float32, device=img.device)
        gt = torch.empty(N, H, W, 3, dtype=torch.float32, device=img.device) if ctx.needs_input_grad[1] else None
        keep, refs = _views(img, target)
        _lib.check(_lib.load().nvdr_image_loss_mean_bwd(*refs, _LOSS.get(ctx.loss, 0), int(ctx.tonemapper == 'log_srgb'), _lib.ptr(dout),
                                                        _lib.ptr(gi), _lib.ptr(gt) if gt is not None else None, _lib.stream_ptr()), 'image_loss_mean_bwd')
        return gi, gt, None, None


def image_loss_mean(img, target, loss='l1', tonemapper='none'):
    """image_loss with the mean folded into the kernels (additive; same value up to the order of the final sum)."""
    return _finite(_image_loss_mean_func.apply(img, target, loss, tonemapper), 'image_loss')


def image_loss(img, target, loss='l1', tonemapper='none', use_python=False):
    '''HDR image loss, tonemapping and loss fused in one kernel; loss in ['l1','mse','smape','relmse','n2n'],
    tonemapper in ['none','log_srgb'] -> scalar.'''
    if use_python:
        out = torch_ref.image_loss(img, target, loss, tonemapper)
    else:
        out = _image_loss_func.apply(img, target, loss, tonemapper)
        out = torch.sum(out) / (img.shape[0] * img.shape[1] * img.shape[2])
    return _finite(out, 'image_loss')


# ----------------------------------------------------------------------------------------------
# transform points (ops.py:503-551)

class _xfm_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, matrix, isPoints):
        _lib.require_cuda_f32(points, 'points')
        _lib.require_cuda_f32(matrix, 'matrix')
        if points.dim() != 3 or points.shape[2] != 3:
            raise RuntimeError("points must have shape [>0, >0, 3] (got %s)" % (tuple(points.shape),))
        if matrix.dim() != 3 or matrix.shape[1] != 4 or matrix.shape[2] != 4:
            raise RuntimeError("matrix must have shape [>0, 4, 4] (got %s)" % (tuple(matrix.shape),))
        ctx.save_for_backward(points, matrix)
        ctx.isPoints = isPoints
        p, m = points.contiguous(), matrix.contiguous()
        B, V = m.shape[0], p.shape[1]
        out = torch.empty(B, V, 4 if isPoints else 3, dtype=torch.float32, device=p.device)
        _lib.check(_lib.load().nvdr_xfm_fwd(_lib.ptr(p), p.shape[0], V, _lib.ptr(m), B, int(bool(isPoints)), _lib.ptr(out),
                                            _lib.stream_ptr()), 'xfm_fwd')
        return out

    @staticmethod
    def backward(ctx, dout):
        points, matrix = ctx.saved_tensors
        m, d = matrix.contiguous(), dout.contiguous()
        B, V = m.shape[0], points.shape[1]
        g = torch.empty(B, V, 3, dtype=torch.float32, device=d.device)
        _lib.check(_lib.load().nvdr_xfm_bwd(_lib.ptr(m), B, V, _lib.ptr(d), int(bool(ctx.isPoints)), _lib.ptr(g),
                                            _lib.stream_ptr()), 'xfm_bwd')
        return g, None, None


def xfm_points(points, matrix, use_python=False):
    '''points [B|1,V,3] x matrix [B,4,4] -> homogeneous [B,V,4].'''
    if use_python:
        out = torch.matmul(torch.nn.functional.pad(points, pad=(0, 1), mode='constant', value=1.0), torch.transpose(matrix, 1, 2))
    else:
        out = _xfm_func.apply(points, matrix, True)
    return _finite(out, 'xfm_points')


def xfm_vectors(vectors, matrix, use_python=False):
    '''vectors [B|1,V,3] x matrix [B,4,4] -> [B,V,3] (no translation).'''
    if use_python:
        out = torch.matmul(torch.nn.functional.pad(vectors, pad=(0, 1), mode='constant', value=0.0),
                           torch.transpose(matrix, 1, 2))[..., 0:3].contiguous()
    else:
        out = _xfm_func.apply(vectors, matrix, False)
    return _finite(out, 'xfm_vectors')


# ----------------------------------------------------------------------------------------------
# fused shading composite (additive: the reference writes this in torch, render/render.py:119-127)

class _shade_composite_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, diff, spec, kd, ks, bsdf):
        for t, name in ((diff, 'diff'), (spec, 'spec')):
            _lib.require_cuda_f32(t, name)
            if t.dim() != 4 or t.shape[3] not in (3, 4):
                raise RuntimeError("shade_composite: %s must be [N,H,W,3] or [N,H,W,4] (got %s)" % (name, tuple(t.shape)))
        _check4(kd, 'shade_composite kd', 3)
        _check4(ks, 'shade_composite ks', 3)
        N, H, W = _extent(diff, spec, kd, ks)
        out = torch.empty(N, H, W, 3, dtype=torch.float32, device=diff.device)
        keep, refs = _views(diff, spec, kd, ks)
        _lib.check(_lib.load().nvdr_shade_composite_fwd(*refs, bsdf, _lib.ptr(out), _lib.stream_ptr()), 'shade_composite_fwd')
        ctx.save_for_backward(diff, spec, kd, ks)
        ctx.bsdf = bsdf
        return out

    @staticmethod
    def backward(ctx, dout):
        diff, spec, kd, ks = ctx.saved_tensors
        N, H, W = _extent(diff, spec, kd, ks)
        dout = dout.contiguous()
        grads = [torch.empty(N, H, W, c, dtype=torch.float32, device=dout.device) for c in (diff.shape[3], spec.shape[3], 3, 3)]
        keep, refs = _views(diff, spec, kd, ks, dout)
        _lib.check(_lib.load().nvdr_shade_composite_bwd(*refs[:4], ctx.bsdf, refs[4], *[_lib.ptr(g) for g in grads],
                                                        _lib.stream_ptr()), 'shade_composite_bwd')
        # broadcast inputs get their gradient at the full extent: fold it (the caller-sums rule of tensor.h:60-62)
        out = []
        for g, t in zip(grads, (diff, spec, kd, ks)):
            for d in range(4):
                if t.shape[d] == 1 and g.shape[d] != 1:
                    g = g.sum(d, keepdim=True)
            out.append(g)
        return tuple(out) + (None,)


class _shade_loss_fused_func(torch.autograd.Function):
    """Composite + mean image loss with their adjoints in ONE launch (csrc/renderutils.hip shade_loss_fused_kernel): the forward already knows
    the upstream gradient of the mean (`upstream`, a device scalar the caller will hand to backward()) and writes the four input gradients."""

    @staticmethod
    def forward(ctx, diff, spec, kd, ks, target, upstream, bsdf, loss, tonemapper):
        for t, name in ((diff, 'diff'), (spec, 'spec')):
            _lib.require_cuda_f32(t, name)
            if t.dim() != 4 or t.shape[3] not in (3, 4):
                raise RuntimeError("shade_loss: %s must be [N,H,W,3] or [N,H,W,4] (got %s)" % (name, tuple(t.shape)))
        _check4(kd, 'shade_loss kd', 3)
        _check4(ks, 'shade_loss ks', 3)
        _check4(target, 'shade_loss target', 3)
        _lib.require_cuda_f32(upstream, 'upstream')
        N, H, W = _extent(diff, spec, kd, ks, target)
        if tuple(kd.shape[:3]) != (N, H, W) or tuple(ks.shape[:3]) != (N, H, W):
            raise RuntimeError('shade_loss: kd / ks must have the full extent')
        lib = _lib.load()
        part = torch.empty(lib.nvdr_image_loss_num_partials(N, H, W) + 1, dtype=torch.float32, device=diff.device)
        grads = [torch.empty(N, H, W, c, dtype=torch.float32, device=diff.device) for c in (diff.shape[3], spec.shape[3], 3, 3)]
        keep, refs = _views(diff, spec, kd, ks, target)
        up = upstream.contiguous().view(1)
        _lib.check(lib.nvdr_shade_loss_fused(*refs[:4], bsdf, refs[4], _LOSS.get(loss, 0), int(tonemapper == 'log_srgb'), _lib.ptr(up), _lib.ptr(part[1:]),
                                             _lib.ptr(part), *[_lib.ptr(g) for g in grads], _lib.stream_ptr()), 'shade_loss_fused')
        ctx.grads = grads
        ctx.upstream = up
        return part[0]

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.grads
        if dout.data_ptr() != ctx.upstream.data_ptr():
            # another upstream gradient than the one the forward was told: rescale (the gradients are linear in it)
            f = dout.reshape(()) / ctx.upstream.reshape(())
            grads = [g * f for g in grads]
        return grads[0], grads[1], grads[2], grads[3], None, None, None, None, None


def shade_composite_loss(diffuse_accum, specular_accum, kd, ks, target, upstream, bsdf='pbr', loss='l1', tonemapper='none'):
    """image_loss_mean(shade_composite(diffuse_accum, specular_accum, kd, ks, bsdf), target, loss, tonemapper) as ONE launch forward + backward (+ the
    fixed-order reduction of the partial sums): the same values and gradients bit for bit.  `upstream`: the device scalar that will be passed to
    `.backward(gradient=...)` (a resident tensor of ones in the training harness); target carries no gradient."""
    return _finite(_shade_loss_fused_func.apply(diffuse_accum, specular_accum, kd, ks, target, upstream, 0 if bsdf == 'pbr' else 1, loss, tonemapper), 'image_loss')


def shade_composite(diffuse_accum, specular_accum, kd, ks, bsdf='pbr', use_python=False):
    '''Final colour of the direct-lighting pass: (diffuse / w) * kd * (1 - metalness) + specular / w for 'pbr',
    (diffuse / w) * kd for 'diffuse' / 'white' (render.py:119-127).  diffuse_accum / specular_accum are either the
    [N,H,W,4] (colour sum, weight) output of the bilateral filter kernel -- the division of ops.py:139-141 is folded
    in -- or plain [N,H,W,3] images.  One kernel per direction instead of ~10 torch kernels.'''
    if use_python:
        out = torch_ref.shade_composite(diffuse_accum, specular_accum, kd, ks, bsdf)
    else:
        out = _shade_composite_func.apply(diffuse_accum, specular_accum, kd, ks, 0 if bsdf == 'pbr' else 1)
    return _finite(out, 'shade_composite')
