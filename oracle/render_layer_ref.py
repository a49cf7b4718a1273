"""Torch restatement of the attribute-interpolation half of render_layer (render/render.py:208-234) -- TEST INFRASTRUCTURE.

The reference takes `rast` / `rast_db` from nvdiffrast's rasteriser and interpolates with `dr.interpolate`.  nvdiffrast is
CUDA/OpenGL only, so its two documented semantics are restated here in plain torch (runs on the CPU):

  dr.interpolate(attr[1,V,C], rast, tri)             attr_pix = u a0 + v a1 + (1 - u - v) a2 with (u, v) = rast[..., 0:2]
                                                     and the triangle rast[..., 3] - 1; zeros where rast[..., 3] == 0
  ... with rast_db and diff_attrs='all'              derivs [.., 2C] INTERLEAVED per attribute: (dA0/dX, dA0/dY, dA1/dX, ...),
                                                     dA/dX = du/dX (a0 - a2) + dv/dX (a1 - a2)

gbuffer_from_rast then follows render.py:208-234 line by line (including `clip_pos_deriv[..., 2:3]` / `[..., 3:4]`, which in
the interleaved layout are the Y-clip derivatives).  tests/test_gpu_gbuffer.py compares nvdr_render_gbuffer with it.
"""
import torch


def interpolate(attr, rast, tri, rast_db=None):
    """attr [V,C], rast [N,H,W,4], tri int [T,3] -> ([N,H,W,C], derivs [N,H,W,2C] or None)."""
    tid = rast[..., 3].long() - 1
    valid = (tid >= 0)[..., None].to(attr.dtype)
    t = tri.long()[tid.clamp(min=0)]
    a0, a1, a2 = attr[t[..., 0]], attr[t[..., 1]], attr[t[..., 2]]
    u, v = rast[..., 0:1], rast[..., 1:2]
    out = (u * a0 + v * a1 + (1.0 - u - v) * a2) * valid
    if rast_db is None:
        return out, None
    dudx, dudy, dvdx, dvdy = (rast_db[..., k:k + 1] for k in range(4))
    dx = dudx * (a0 - a2) + dvdx * (a1 - a2)
    dy = dudy * (a0 - a2) + dvdy * (a1 - a2)
    d = torch.stack([dx, dy], dim=-1).reshape(*dx.shape[:-1], -1) * valid          # (dA0/dX, dA0/dY, dA1/dX, ...)
    return out, d


def _safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))       # render/util.py:27-31


def gbuffer_from_rast(mesh, rast, rast_db, v_pos_clip):
    """mesh: dict of CPU tensors; rast, rast_db [N,H,W,4]; v_pos_clip [N,V,4] (render.py:271).  Returns the tensors
    render_layer hands to shade(): gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_texc, gb_texc_deriv, gb_depth."""
    out = {}
    out['gb_pos'], _ = interpolate(mesh['v_pos'], rast, mesh['t_pos_idx'])                                     # render.py:208
    v0, v1, v2 = (mesh['v_pos'][mesh['t_pos_idx'][:, k].long()] for k in range(3))
    face_normals = _safe_normalize(torch.cross(v1 - v0, v2 - v0, dim=-1))                                       # render.py:211-214
    fidx = torch.arange(face_normals.shape[0])[:, None].repeat(1, 3)
    out['gb_geometric_normal'], _ = interpolate(face_normals, rast, fidx)                                      # render.py:216
    out['gb_normal'], _ = interpolate(mesh['v_nrm'], rast, mesh['t_nrm_idx'])                                  # render.py:220
    out['gb_tangent'], _ = interpolate(mesh['v_tng'], rast, mesh['t_tng_idx'])                                 # render.py:221
    out['gb_texc'], out['gb_texc_deriv'] = interpolate(mesh['v_tex'], rast, mesh['t_tex_idx'], rast_db)       # render.py:225
    eps = 0.00001
    depth = []
    for n in range(rast.shape[0]):                                                                             # render.py:228-234
        clip_pos, clip_pos_deriv = interpolate(v_pos_clip[n], rast[n:n + 1], mesh['t_pos_idx'], rast_db[n:n + 1])
        z0 = torch.clamp(clip_pos[..., 2:3], min=eps) / torch.clamp(clip_pos[..., 3:4], min=eps)
        z1 = torch.clamp(clip_pos[..., 2:3] + torch.abs(clip_pos_deriv[..., 2:3]), min=eps) / \
            torch.clamp(clip_pos[..., 3:4] + torch.abs(clip_pos_deriv[..., 3:4]), min=eps)
        depth.append(torch.cat((z0, torch.abs(z1 - z0)), dim=-1))
    out['gb_depth'] = torch.cat(depth, dim=0)
    return out
