"""Deterministic benchmark / test scenes for the direct-lighting path.

Nothing here is on the hot path: it builds the INPUTS the reference gets from nvdiffrast,
DatasetMesh and the HDR probes (none of which exist on ROCm / in this checkout):

  * meshes        : assets/{bob,spot}.npz (CC0), vertex normals as render/mesh.py:150-178; assets/dmtet64_{init,mid}.npz: marching-tets
                    extractions from the reference's tet grid with seeded SDFs (tools/make_dmtet_mesh.py)
  * cameras       : DatasetMesh._rotate_scene (dataset/dataset_mesh.py:62-71) with the matrices
                    of render/util.py:185-210 (perspective 45 deg, radius 3, rotate_x(-0.4))
  * env maps      : E0 uniform 0.5 (the trainable init, train.py:612) and E1 seeded "sky + suns"
  * permutations  : argsort(rand(32768, S)) as render/optixutils/ops.py:84-86, but from a seeded
                    CPU generator so that the CPU oracle and the GPU see the same table
  * G-buffer      : attribute interpolation from primary-ray hits (the hits come from
                    nvdr_trace_closest on the GPU, or from the oracle's brute force on the CPU)

All functions are pure torch and device-agnostic.
"""
import math
import os

import numpy as np
import torch

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'assets')


def dmtet_atlas(face_gidx, n_tets):
    """Texture coordinates of a marching-tets mesh as the reference lays them out (geometry/dmtet.py:50-79 map_uv, called with
    max_idx = 2 * num_tets): every tet owns one cell of an N x N atlas, N = ceil(sqrt(num_tets)); its (up to) two triangles are the two
    halves of the cell's quad, shrunk by pad = 0.9 / N.  Returns (v_tex [4 N^2, 2] float32, t_tex_idx [F, 3] int32)."""
    N = int(math.ceil(math.sqrt((2 * int(n_tets) + 1) // 2)))
    lin = torch.linspace(0, 1 - (1 / N), N, dtype=torch.float32).numpy()      # (torch's float32 linspace, as the reference: numpy's rounds differently)
    tex_y, tex_x = np.meshgrid(lin, lin, indexing='ij')
    pad = np.float32(0.9 / N)
    uvs = np.stack([tex_x, tex_y, tex_x + pad, tex_y, tex_x + pad, tex_y + pad, tex_x, tex_y + pad], -1).reshape(-1, 2).astype(np.float32)
    g = np.asarray(face_gidx, dtype=np.int64)
    tet, tri = g // 2, g % 2
    idx = np.stack((tet * 4, tet * 4 + tri + 1, tet * 4 + tri + 2), -1).astype(np.int32)
    return torch.from_numpy(uvs), torch.from_numpy(idx)


def _procedural_kd(res=512, seed=4321):
    """A smooth seeded albedo for the meshes that come without one (the DMTet extractions): low-frequency colour blobs in [0.15, 0.85]."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid((np.arange(res) + 0.5) / res, (np.arange(res) + 0.5) / res, indexing='ij')
    img = np.full((res, res, 3), 0.5)
    for _ in range(12):
        cx, cy, sig = rng.random(), rng.random(), 0.08 + 0.2 * rng.random()
        col = rng.random(3) - 0.5
        img += 0.6 * col * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sig ** 2))[..., None]
    return torch.from_numpy(np.clip(img, 0.15, 0.85).astype(np.float32))


def load_mesh(name='bob', device='cpu'):
    d = np.load(os.path.join(_ASSETS, name + '.npz'))
    if 'face_gidx' in d.files:
        # a marching-tets extraction (tools/make_dmtet_mesh.py): positions + triangles + the per-face global index the atlas is rebuilt from
        m = {'v_pos': torch.from_numpy(d['v_pos']), 't_pos_idx': torch.from_numpy(d['t_pos_idx'])}
        m['v_tex'], m['t_tex_idx'] = dmtet_atlas(d['face_gidx'], int(d['n_tets']))
        m['ks'] = torch.tensor([0.0, 0.25, 0.0])
        m['kd_tex'] = _procedural_kd()
        m = {k: v.to(device) for k, v in m.items()}
        m['name'] = name
        return finish_mesh(m)
    m = {k: torch.from_numpy(d[k]) for k in ('v_pos', 'v_tex', 't_pos_idx', 't_tex_idx', 'ks')}
    m['kd_tex'] = torch.from_numpy(d['kd_tex'].astype(np.float32))
    m = {k: v.to(device) for k, v in m.items()}
    m['name'] = name
    return finish_mesh(m)


def finish_mesh(m):
    """Vertex normals (render/mesh.py:150-178: one per position vertex, so t_nrm_idx = t_pos_idx) and tangents
    (render/mesh.py:181-219, t_tng_idx = t_nrm_idx) of a mesh dict with v_pos, t_pos_idx, v_tex, t_tex_idx."""
    m['v_nrm'] = auto_normals(m['v_pos'], m['t_pos_idx'])
    m['t_nrm_idx'] = m['t_pos_idx']
    m['v_tng'] = compute_tangents(m)
    m['t_tng_idx'] = m['t_nrm_idx']
    return m


def _safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))       # render/util.py:27-31


def compute_tangents(m):
    """Per-vertex tangents, behaviour of render/mesh.py:181-219: per-triangle tangent from the uv parametrisation, averaged
    over the triangles of a (normal-indexed) vertex, Gram-Schmidt against the vertex normal."""
    tp, tt, tn = m['t_pos_idx'].long(), m['t_tex_idx'].long(), m['t_nrm_idx'].long()
    pos = [m['v_pos'][tp[:, i]] for i in range(3)]
    tex = [m['v_tex'][tt[:, i]] for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]
    denom = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]
    tang = nom / torch.where(denom > 0.0, torch.clamp(denom, min=1e-6), torch.clamp(denom, max=-1e-6))
    tangents, tansum = torch.zeros_like(m['v_nrm']), torch.zeros_like(m['v_nrm'])
    for i in range(3):
        idx = tn[:, i:i + 1].repeat(1, 3)
        tangents.scatter_add_(0, idx, tang)
        tansum.scatter_add_(0, idx, torch.ones_like(tang))
    tangents = tangents / tansum.clamp(min=1.0)          # (a vertex no triangle references keeps a zero tangent instead of 0/0)
    tangents = _safe_normalize(tangents)
    return _safe_normalize(tangents - (tangents * m['v_nrm']).sum(-1, keepdim=True) * m['v_nrm']).contiguous()


def subdivide_mesh(m, levels, cache_key=None):
    """Midpoint subdivision of positions AND texture coordinates (each in its own index space: every triangle splits into the
    same four children in both, so the rows of t_pos_idx and t_tex_idx stay aligned), then fresh normals / tangents.
    With NVDR_MESH_CACHE=<dir> and a cache_key the result is kept on disk (fresh-process surveys: 4 s per process saved)."""
    import os
    cache = os.environ.get('NVDR_MESH_CACHE')
    path = os.path.join(cache, 'nvdr_mesh_%s_s%d.pt' % (cache_key, levels)) if (cache and cache_key) else None
    if path and os.path.exists(path):
        return torch.load(path)
    out = _subdivide_mesh(m, levels)
    if path:
        tmp = '%s.%d.tmp' % (path, os.getpid())
        torch.save(out, tmp)
        os.replace(tmp, path)
    return out


def _subdivide_mesh(m, levels):
    m = dict(m)
    m['v_pos'], m['t_pos_idx'] = subdivide(m['v_pos'], m['t_pos_idx'], levels)
    vt3 = torch.cat([m['v_tex'], torch.zeros_like(m['v_tex'][:, :1])], dim=-1)
    vt3, m['t_tex_idx'] = subdivide(vt3, m['t_tex_idx'], levels)
    m['v_tex'] = vt3[:, :2].contiguous()
    return finish_mesh(m)


def auto_normals(v_pos, t_pos_idx):
    """Area-weighted vertex normals (behaviour of render/mesh.py:150-178)."""
    i0, i1, i2 = (t_pos_idx[:, k].long() for k in range(3))
    v0, v1, v2 = v_pos[i0], v_pos[i1], v_pos[i2]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)
    vn = torch.zeros_like(v_pos)
    for idx in (i0, i1, i2):
        vn.scatter_add_(0, idx[:, None].expand(-1, 3), fn)
    dot = (vn * vn).sum(-1, keepdim=True)
    vn = torch.where(dot > 1e-20, vn, torch.tensor([0.0, 0.0, 1.0], dtype=vn.dtype, device=vn.device))
    return torch.nn.functional.normalize(vn, dim=-1)


def subdivide(v_pos, t_pos_idx, levels=1):
    """Midpoint (1-to-4) subdivision: a stand-in for DMTet-sized meshes (config 4/5 of BASELINE.json)."""
    for _ in range(levels):
        t = t_pos_idx.long()
        e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], 0)
        e_sorted, _ = torch.sort(e, dim=1)
        uniq, inv = torch.unique(e_sorted, dim=0, return_inverse=True)
        mid = 0.5 * (v_pos[uniq[:, 0]] + v_pos[uniq[:, 1]])
        nv = v_pos.shape[0]
        T = t.shape[0]
        m01, m12, m20 = inv[:T] + nv, inv[T:2 * T] + nv, inv[2 * T:] + nv
        v_pos = torch.cat([v_pos, mid], 0)
        t_pos_idx = torch.cat([torch.stack([t[:, 0], m01, m20], 1), torch.stack([t[:, 1], m12, m01], 1),
                               torch.stack([t[:, 2], m20, m12], 1), torch.stack([m01, m12, m20], 1)], 0).int()
    return v_pos.contiguous(), t_pos_idx.contiguous()


# ----------------------------------------------------------------------------------------------
# cameras

def perspective(fovy=math.radians(45.0), aspect=1.0, n=0.1, f=1000.0):
    y = math.tan(fovy / 2)
    return torch.tensor([[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0],
                         [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]], dtype=torch.float32)


def translate(x, y, z):
    return torch.tensor([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=torch.float32)


def rotate_x(a):
    s, c = math.sin(a), math.cos(a)
    return torch.tensor([[1, 0, 0, 0], [0, c, s, 0], [0, -s, c, 0], [0, 0, 0, 1]], dtype=torch.float32)


def rotate_y(a):
    s, c = math.sin(a), math.cos(a)
    return torch.tensor([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=torch.float32)


def camera(k, n_views, radius=3.0, aspect=1.0):
    """View k of n_views on the validation orbit: returns (mv, mvp, campos)."""
    ang = (k / n_views) * math.pi * 2
    mv = translate(0, 0, -radius) @ (rotate_x(-0.4) @ rotate_y(ang))
    mvp = perspective(aspect=aspect) @ mv
    campos = torch.linalg.inv(mv)[:3, 3]
    return mv, mvp, campos


def camera_rays(mv, aspect=1.0, fovy=math.radians(45.0)):
    """[4,3] = (eye, U, V, W): the primary ray through NDC (X, Y) of the view `mv` is normalize(X U + Y V + W) from eye --
    the pinhole camera perspective() projects with (Y flipped as in render/util.py:185-194)."""
    yt = math.tan(fovy / 2)
    rot = mv[:3, :3]
    eye = torch.linalg.inv(mv)[:3, 3]
    return torch.stack([eye, yt * aspect * rot[0], -yt * rot[1], -rot[2]]).contiguous()


def primary_rays(mv, H, W, fovy=math.radians(45.0)):
    """World-space primary rays through the pixel centres (row 0 = clip y -1, as the rasteriser lays out)."""
    yt = math.tan(fovy / 2)
    aspect = W / H
    py, px = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    cx = (px + 0.5) / W * 2 - 1
    cy = (py + 0.5) / H * 2 - 1
    d_view = torch.stack([cx * yt * aspect, -cy * yt, -torch.ones_like(cx)], -1)
    rot = mv[:3, :3]
    d_world = torch.nn.functional.normalize(d_view @ rot, dim=-1)  # rot^T applied to row vectors
    campos = torch.linalg.inv(mv)[:3, 3]
    return campos.expand(H, W, 3).contiguous(), d_world.contiguous()


# ----------------------------------------------------------------------------------------------
# G-buffer from primary hits

def gbuffer_from_hits(mesh, hit_t, hit_tri, hit_uv, ray_o, ray_d, kd_mode='texture'):
    """hit_*: [H,W], [H,W] int, [H,W,2] -> dict of [1,H,W,C] float32 tensors (what render_layer produces,
    render/render.py:208-234, minus derivatives): mask, gb_pos, gb_normal (smooth), gb_geometric_normal, kd, ks, depth."""
    dev = hit_t.device
    H, W = hit_t.shape
    mask = (hit_tri >= 0)
    tri = hit_tri.clamp(min=0).long()
    u, v = hit_uv[..., 0:1], hit_uv[..., 1:2]
    w0 = 1.0 - u - v
    t = mesh['t_pos_idx'].long()[tri]
    vp, vn = mesh['v_pos'], mesh['v_nrm']
    p0, p1, p2 = vp[t[..., 0]], vp[t[..., 1]], vp[t[..., 2]]
    pos = w0 * p0 + u * p1 + v * p2
    nrm = torch.nn.functional.normalize(w0 * vn[t[..., 0]] + u * vn[t[..., 1]] + v * vn[t[..., 2]], dim=-1)
    gnrm = torch.nn.functional.normalize(torch.cross(p1 - p0, p2 - p0, dim=-1), dim=-1)
    if kd_mode == 'texture':
        tt = mesh['t_tex_idx'].long()[tri]
        vt = mesh['v_tex']
        uv = w0 * vt[tt[..., 0]] + u * vt[tt[..., 1]] + v * vt[tt[..., 2]]
        tex = mesh['kd_tex']
        R = tex.shape[0]
        ix = (uv[..., 0] * R).long().clamp(0, R - 1)
        iy = ((1.0 - uv[..., 1]) * R).long().clamp(0, R - 1)
        kd = tex[iy, ix]
    else:
        kd = torch.full((H, W, 3), 0.5, device=dev)
    ks = mesh['ks'].to(dev).expand(H, W, 3)
    m = mask[..., None].float()
    z = (hit_t * m[..., 0])
    out = {
        'mask': m[None, ..., 0].contiguous(),
        'gb_pos': (pos * m)[None].contiguous(),
        'gb_normal': (nrm * m)[None].contiguous(),
        'gb_geometric_normal': (gnrm * m)[None].contiguous(),
        'kd': (kd * m)[None].contiguous(),
        'ks': (ks * m)[None].contiguous(),
        'depth': z[None, ..., None].contiguous(),
    }
    return out


# ----------------------------------------------------------------------------------------------
# environment light

def env_map(kind='E1', res=256):
    """E0: uniform 0.5.  E1: vertical gradient + 3 seeded Gaussian 'suns' (SURVEY 8d), clamp min 1e-4."""
    H = W = res
    if kind == 'E0':
        return torch.full((H, W, 3), 0.5, dtype=torch.float32)
    rng = np.random.default_rng(1234)
    y, x = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing='ij')
    base = (0.2 + 0.3 * y)[..., None] * np.ones(3)
    for _ in range(3):
        cx, cy = rng.random(), 0.15 + 0.5 * rng.random()
        sig = (2.0 + 2.0 * rng.random()) / res
        peak = 200.0 + 1800.0 * rng.random()
        col = 0.7 + 0.3 * rng.random(3)
        dx = np.minimum(np.abs(x - cx), 1 - np.abs(x - cx))
        g = np.exp(-(dx ** 2 + (y - cy) ** 2) / (2 * sig ** 2))
        base = base + peak * g[..., None] * col
    return torch.from_numpy(np.maximum(base, 1e-4).astype(np.float32))


def light_tables(base):
    """pdf / cols / rows of EnvironmentLight.update_pdf (render/light.py:46-59), in torch on base's device.
    Returns (pdf [H,W], rows [H,W] with identical columns, cols [H,W])."""
    H, W = base.shape[0], base.shape[1]
    Y = ((torch.arange(H, dtype=torch.float32, device=base.device) + 0.5) / H)[:, None].expand(H, W)
    pdf = torch.max(base, dim=-1)[0] * torch.sin(Y * np.pi)
    pdf = pdf / torch.sum(pdf)
    cols = torch.cumsum(pdf, dim=1)
    rows = torch.cumsum(cols[:, -1:].repeat([1, W]), dim=0)
    cols = cols / torch.where(cols[:, -1:] > 0, cols[:, -1:], torch.ones_like(cols))
    rows = rows / torch.where(rows[-1:, :] > 0, rows[-1:, :], torch.ones_like(rows))
    return pdf.contiguous(), rows.contiguous(), cols.contiguous()


def perms_table(n_samples_x, seed=0, n_perms=32768):
    g = torch.Generator(device='cpu').manual_seed(seed)
    S = n_samples_x * n_samples_x
    return torch.argsort(torch.rand(n_perms, S, generator=g), dim=-1).int().contiguous()
