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


# ----------------------------------------------------------------------------------------------
# The gradient route from the G-buffer to the trained vertices and textures (round 4): torch restatements that
# AUTOGRAD differentiates on the CPU -- the checker of csrc/mesh.hip (nvdr_mesh_frame_*, nvdr_interpolate_bwd,
# nvdr_texture_lookup_*).  Double precision when the inputs are.

def auto_normals(v_pos, t_pos_idx):
    """render/mesh.py:150-178: area-weighted vertex normals by scatter_add_, degenerate ones replaced by (0, 0, 1)."""
    t = t_pos_idx.long()
    i0, i1, i2 = t[:, 0], t[:, 1], t[:, 2]
    v0, v1, v2 = v_pos[i0], v_pos[i1], v_pos[i2]
    face_normals = torch.cross(v1 - v0, v2 - v0, dim=-1)                                          # mesh.py:160
    v_nrm = torch.zeros_like(v_pos)
    v_nrm = v_nrm.scatter_add(0, i0[:, None].repeat(1, 3), face_normals)                           # mesh.py:163-166
    v_nrm = v_nrm.scatter_add(0, i1[:, None].repeat(1, 3), face_normals)
    v_nrm = v_nrm.scatter_add(0, i2[:, None].repeat(1, 3), face_normals)
    dflt = torch.tensor([0.0, 0.0, 1.0], dtype=v_pos.dtype)
    v_nrm = torch.where((v_nrm * v_nrm).sum(-1, keepdim=True) > 1e-20, v_nrm, dflt)                # mesh.py:169
    return _safe_normalize(v_nrm)                                                                  # mesh.py:170


def compute_tangents(v_pos, v_nrm, v_tex, t_pos_idx, t_tex_idx):
    """render/mesh.py:181-219 with t_nrm_idx = t_pos_idx (mesh.py:178)."""
    tp, tt = t_pos_idx.long(), t_tex_idx.long()
    pos = [v_pos[tp[:, i]] for i in range(3)]
    tex = [v_tex[tt[:, i]].to(v_pos.dtype) for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]                                              # mesh.py:199
    denom = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]                      # mesh.py:200
    tang = nom / torch.where(denom > 0.0, torch.clamp(denom, min=1e-6), torch.clamp(denom, max=-1e-6))   # mesh.py:203
    tangents, tansum = torch.zeros_like(v_nrm), torch.zeros_like(v_nrm)
    for i in range(3):                                                                             # mesh.py:206-209
        idx = tp[:, i:i + 1].repeat(1, 3)
        tangents = tangents.scatter_add(0, idx, tang)
        tansum = tansum.scatter_add(0, idx, torch.ones_like(tang))
    tangents = tangents / tansum.clamp(min=1.0)      # (the reference divides by tansum: 0 / 0 for a vertex no triangle references)
    tangents = _safe_normalize(tangents)                                                           # mesh.py:213
    return _safe_normalize(tangents - (tangents * v_nrm).sum(-1, keepdim=True) * v_nrm)            # mesh.py:214


def ray_barycentrics(v_pos, t_pos_idx, rast, cam):
    """(u, v) [N,H,W,1] each of the FIXED primary ray of every covered pixel as a differentiable function of the vertices
    (Moeller-Trumbore in nvdiffrast's convention attr = u a0 + v a1 + (1 - u - v) a2): the function nvdiffrast's rasterize
    backward differentiates (barycentrics are invariant under the projective map to clip space).  cam [N,4,3] = (eye, U, V, W)."""
    N, H, W = rast.shape[:3]
    tid = rast[..., 3].long() - 1
    t = t_pos_idx.long()[tid.clamp(min=0)]
    p0, p1, p2 = v_pos[t[..., 0]], v_pos[t[..., 1]], v_pos[t[..., 2]]
    dt = v_pos.dtype
    # the pixel's ray exactly as the G-buffer kernel forms it (fp32, same association): the FIXED ray is an input, not a result
    X = ((torch.arange(W, dtype=torch.float32) + 0.5) / W * 2 - 1)[None, None, :, None]
    Y = ((torch.arange(H, dtype=torch.float32) + 0.5) / H * 2 - 1)[None, :, None, None]
    cam = cam.float()
    d = ((cam[:, None, None, 1] * X + cam[:, None, None, 2] * Y) + cam[:, None, None, 3]).to(dt)
    eye = cam[:, None, None, 0].expand(N, H, W, 3).to(dt)
    e1, e2, s = p1 - p0, p2 - p0, eye - p0
    pv = torch.cross(d, e2, dim=-1)
    det = (e1 * pv).sum(-1, keepdim=True)
    det = torch.where(det.abs() > 1e-30, det, torch.ones_like(det))
    b1 = (s * pv).sum(-1, keepdim=True) / det
    b2 = (d * torch.cross(s, e1, dim=-1)).sum(-1, keepdim=True) / det
    return 1.0 - b1 - b2, b1


def gbuffer_differentiable(v_pos, v_nrm, v_tng, t_pos_idx, rast, cam=None, ray_values=False):
    """gb_pos, gb_geometric_normal, gb_normal, gb_tangent of render.py:208-222 for the coverage in `rast`, differentiable w.r.t.
    v_pos / v_nrm / v_tng; with cam the barycentrics follow the vertices too, otherwise rast's (u, v) are constants."""
    tid = rast[..., 3].long() - 1
    valid = (tid >= 0)[..., None].to(v_pos.dtype)
    t = t_pos_idx.long()[tid.clamp(min=0)]
    u, v = rast[..., 0:1].to(v_pos.dtype), rast[..., 1:2].to(v_pos.dtype)
    if cam is not None:
        # VALUES from the rasteriser's output (what the forward pass interpolated with), DERIVATIVES from the ray formulation:
        # the two agree to ~1e-4 in value (tests/test_gpu_geometry.py checks that), and the adjoint under test is defined on rast's values
        # (ray_values=True takes the values from the ray formulation too: finite-difference checks of the barycentric term)
        ur, vr = ray_barycentrics(v_pos, t_pos_idx, rast, cam)
        u, v = (ur, vr) if ray_values else (u + (ur - ur.detach()), v + (vr - vr.detach()))

    def interp(attr):
        return (u * attr[t[..., 0]] + v * attr[t[..., 1]] + (1.0 - u - v) * attr[t[..., 2]]) * valid
    p0, p1, p2 = v_pos[t[..., 0]], v_pos[t[..., 1]], v_pos[t[..., 2]]
    fn = _safe_normalize(torch.cross(p1 - p0, p2 - p0, dim=-1))                                    # render.py:211-214
    return {'gb_pos': interp(v_pos), 'gb_geometric_normal': (u * fn + v * fn + (1.0 - u - v) * fn) * valid,
            'gb_normal': interp(v_nrm), 'gb_tangent': interp(v_tng)}


def texture_lookup(tex, texc, rast):
    """Nearest texel of tex [R,R,3] at texc [..., 2] (the stand-in for Texture2D.sample, render.py:61-68, that the harness uses):
    (ix, iy) = (clamp(int(s R)), clamp(int((1 - t) R))), zero where rast[..., 3] <= 0.  Differentiable w.r.t. tex."""
    R = tex.shape[0]
    ix = (texc[..., 0] * R).long().clamp(0, R - 1)
    iy = ((1.0 - texc[..., 1]) * R).long().clamp(0, R - 1)
    return tex[iy, ix] * (rast[..., 3:4] > 0).to(tex.dtype)


def shade_composite(diff, spec, kd, ks, bsdf='pbr'):
    """Final colour of shade() from the two accumulated (or filtered) light images: render/render.py:119-127; a 4-channel input is the
    (sum w c, sum w) pair of the bilateral filter, divided as bilateral_denoiser does (render/optixutils/ops.py:139-141)."""
    d = diff[..., 0:3] / diff[..., 3:4] if diff.shape[-1] == 4 else diff
    s = spec[..., 0:3] / spec[..., 3:4] if spec.shape[-1] == 4 else spec
    if bsdf in ('white', 'diffuse'):
        return d * kd
    return d * (kd * (1.0 - ks[..., 2:3])) + s
