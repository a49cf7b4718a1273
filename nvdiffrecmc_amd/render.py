"""The differentiable half of render_layer / shade that sits between the trained parameters and the hot path
(render/render.py:25,61-68,208-234), on MI355X: G-buffer attributes that carry gradients back to the vertices, and the
texel lookups of the trained kd / ks / normal textures.

    gbuffer(ctx, v_pos, v_nrm, v_tng, topo, mvp, cam, (H, W))   rasterize + interpolate (render.py:208-234, :308-310)
    texture_lookup((kd, ks, nrm), gb_texc, rast)                Texture2D.sample of the three material textures (render.py:61-68)

Forward is csrc/gbuffer.hip (primary rays through the context's BVH), backward csrc/mesh.hip: the adjoint of dr.interpolate
plus the barycentric term nvdiffrast's rasterize backward supplies.  dr.antialias (silhouette gradients, render.py:290) and
the trilinear mip filter of dr.texture are NOT reproduced: coverage is not differentiated and the textures are sampled at the
nearest texel.
"""
import ctypes

import torch

from . import _lib
from . import optixutils as ou


class _gbuffer_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, optix_ctx, v_pos, v_nrm, v_tng, topo, mvp, cam, resolution, bary_grad):
        ctx.set_materialize_grads(False)         # (an attribute nobody differentiated arrives as None below, not as an image of zeros)
        mesh = {'v_pos': v_pos.contiguous(), 't_pos_idx': topo.t_pos_idx, 'v_nrm': v_nrm.contiguous(), 't_nrm_idx': topo.t_pos_idx,
                'v_tng': v_tng.contiguous(), 't_tng_idx': topo.t_pos_idx, 'v_tex': topo.v_tex, 't_tex_idx': topo.t_tex_idx}
        gb = ou.render_gbuffer(optix_ctx, mesh, mvp, cam, resolution)
        ctx.save_for_backward(gb['rast'], mesh['v_pos'], mesh['v_nrm'], mesh['v_tng'], cam.contiguous())
        ctx.topo, ctx.bary_grad = topo, bool(bary_grad)
        outs = (gb['rast'], gb['rast_db'], gb['gb_pos'], gb['gb_geometric_normal'], gb['gb_normal'], gb['gb_tangent'], gb['gb_texc'],
                gb['gb_texc_deriv'], gb['gb_depth'])
        # coverage, texture coordinates (v_tex is not trained) and the depth pair (torch.no_grad in render.py:228-234) carry no gradient
        ctx.mark_non_differentiable(gb['rast'], gb['rast_db'], gb['gb_texc'], gb['gb_texc_deriv'], gb['gb_depth'])
        return outs

    @staticmethod
    def backward(ctx, _g_rast, _g_rast_db, g_pos, g_gn, g_nrm, g_tng, _g_texc, _g_texc_db, _g_depth):
        rast, v_pos, v_nrm, v_tng, cam = ctx.saved_tensors
        topo = ctx.topo
        N, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        a = _lib.NvdrInterpolateBwdArgs()
        a.rast, a.n, a.h, a.w = rast.data_ptr(), N, H, W
        a.v_pos, a.t_pos_idx, a.v_nrm, a.v_tng = v_pos.data_ptr(), topo.t_pos_idx.data_ptr(), v_nrm.data_ptr(), v_tng.data_ptr()
        a.n_verts, a.n_tris = topo.n_verts, topo.n_tris
        a.cam = cam.data_ptr() if ctx.bary_grad else None
        keep = []
        for name, g in (('gb_pos_grad', g_pos), ('gb_geometric_normal_grad', g_gn), ('gb_normal_grad', g_nrm), ('gb_tangent_grad', g_tng)):
            if g is not None:
                g = g.contiguous()
                keep.append(g)
                setattr(a, name, g.data_ptr())
        out = torch.zeros(3, topo.n_verts, 3, dtype=torch.float32, device=rast.device)
        a.v_pos_grad, a.v_nrm_grad, a.v_tng_grad = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
        _lib.check(_lib.load().nvdr_interpolate_bwd(ctypes.byref(a), _lib.stream_ptr()), 'interpolate_bwd')
        return None, out[0], out[1], out[2], None, None, None, None, None


_GB_NAMES = ('rast', 'rast_db', 'gb_pos', 'gb_geometric_normal', 'gb_normal', 'gb_tangent', 'gb_texc', 'gb_texc_deriv', 'gb_depth')


def gbuffer(optix_ctx, v_pos, v_nrm, v_tng, topo, mvp, cam, resolution, bary_grad=True):
    """Dict of the G-buffer tensors of N views (see optixutils.render_gbuffer); gb_pos, gb_geometric_normal, gb_normal and
    gb_tangent are differentiable w.r.t. v_pos / v_nrm / v_tng.  The context's BVH must hold (v_pos, topo.t_pos_idx).
    bary_grad=False holds the barycentrics constant (only the attribute values are differentiated)."""
    return dict(zip(_GB_NAMES, _gbuffer_func.apply(optix_ctx, v_pos, v_nrm, v_tng, topo, mvp, cam, resolution, bary_grad)))


class PersistentGrads(list):
    """Persistent [R,R,3] buffers the texture lookup's adjoint ADDS into (texture_lookup(grad_buffers=...)), with the contract made
    explicit: all zero when a backward pass runs, then `dirty` until their consumer has used AND re-zeroed them (FusedAdam zero_grad, or
    a memset) and says so (`dirty = False`).  A second backward pass into dirty buffers would double-count and raises."""

    def __init__(self, buffers):
        super().__init__(buffers)
        for b in self:
            if not b.is_contiguous():
                raise ValueError('PersistentGrads: buffers must be contiguous')
        self.dirty = False


class _texture_lookup_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, texc, rast, grad_buffers, *textures):
        ctx.set_materialize_grads(False)
        ctx.grad_buffers = grad_buffers
        n = len(textures)
        if not 1 <= n <= _lib.MAX_TEXTURES:
            raise RuntimeError('texture_lookup: 1..%d textures' % _lib.MAX_TEXTURES)
        _lib.require_cuda_f32(texc, 'gb_texc')
        _lib.require_cuda_f32(rast, 'rast')
        texc, rast = texc.contiguous(), rast.contiguous()
        P = rast.numel() // 4
        if rast.shape[-1] != 4 or texc.shape[-1] != 2 or texc.numel() != 2 * P:
            raise RuntimeError('texture_lookup: gb_texc %s must hold one (s, t) pair per pixel of rast %s' % (tuple(texc.shape), tuple(rast.shape)))
        a = _lib.NvdrTextureArgs()
        a.n_tex, a.texc, a.rast, a.n_pix = n, texc.data_ptr(), rast.data_ptr(), P
        outs, keep = [], []
        for k, t in enumerate(textures):
            _lib.require_cuda_f32(t, 'texture %d' % k)
            if t.dim() != 3 or t.shape[0] != t.shape[1] or t.shape[2] != 3:
                raise RuntimeError('texture %d must be [R,R,3] (got %s)' % (k, tuple(t.shape)))
            t = t.contiguous()
            keep.append(t)
            o = torch.empty(*rast.shape[:-1], 3, dtype=torch.float32, device=rast.device)
            outs.append(o)
            a.res[k], a.tex[k], a.out[k] = t.shape[0], t.data_ptr(), o.data_ptr()
        _lib.check(_lib.load().nvdr_texture_lookup_fwd(ctypes.byref(a), _lib.stream_ptr()), 'texture_lookup_fwd')
        ctx.save_for_backward(texc, rast)
        ctx.res = [t.shape[0] for t in textures]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        texc, rast = ctx.saved_tensors
        a = _lib.NvdrTextureArgs()
        a.n_tex, a.texc, a.rast, a.n_pix = len(ctx.res), texc.data_ptr(), rast.data_ptr(), rast.numel() // 4
        grads, keep = [], []
        if isinstance(ctx.grad_buffers, PersistentGrads):
            # (under HIP-graph capture this runs once, at capture time; a replay repeats the launches, not the check)
            if ctx.grad_buffers.dirty:
                raise RuntimeError('texture_lookup: the persistent gradient buffers still hold the gradients of an earlier backward pass '
                                   '(two lookups sharing one set of buffers, or a consumer that did not clear them): they would be counted twice')
            ctx.grad_buffers.dirty = True
        for k, (R, g) in enumerate(zip(ctx.res, gouts)):
            g = g.contiguous() if g is not None else torch.zeros(*rast.shape[:-1], 3, dtype=torch.float32, device=rast.device)
            keep.append(g)
            # a caller-owned persistent buffer (all zero on entry: its consumer re-zeroes what it used, FusedAdam zero_grad) or a fresh one
            # (returned as a FRESH view: autograd adopts a gradient without copying only when nothing else references that tensor object)
            d = ctx.grad_buffers[k].view(R, R, 3) if ctx.grad_buffers is not None else torch.empty(R, R, 3, dtype=torch.float32, device=rast.device)
            grads.append(d)
            a.res[k], a.dout[k], a.dtex[k] = R, g.data_ptr(), d.data_ptr()
        a.accumulate = 1 if ctx.grad_buffers is not None else 0
        _lib.check(_lib.load().nvdr_texture_lookup_bwd(ctypes.byref(a), _lib.stream_ptr()), 'texture_lookup_bwd')
        return (None, None, None) + tuple(grads)


def texture_lookup(textures, gb_texc, rast, grad_buffers=None):
    """[N,H,W,3] per texture: its texel nearest to gb_texc where rast[..., 3] > 0, zero elsewhere; one launch for all of them.
    grad_buffers: optional list of persistent [R,R,3] tensors the backward pass ADDS the texture gradients to and returns (they must
    be all zero when backward runs: no memset of 3 x 12.6 MB per iteration when the optimizer zeroes what it consumed)."""
    return _texture_lookup_func.apply(gb_texc, rast, grad_buffers, *textures)
