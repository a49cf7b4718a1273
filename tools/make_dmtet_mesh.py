#!/usr/bin/env python3
"""DMTet-shaped benchmark meshes: marching tetrahedra over the reference's 64^3 tet grid with a seeded SDF.

What DMTetGeometry.getMesh hands optix_build_bvh every iteration (/root/reference/geometry/dmtet.py:185-202) is not a clean
subdivided surface but the zero set of a per-vertex SDF on an irregular (quartet "acute") tet grid: triangles of every shape and
size, slivers where the surface grazes a grid vertex, and -- early in training -- floaters and internal sheets all over the volume.
The subdivided-bob stand-ins of rounds 2-5 are the case a Morton-code BVH handles best; these are not.

The extraction below RESTATES marching_tets (dmtet.py:91-141: sign pattern -> 1 or 2 triangles per crossing tet, vertices on the
crossing edges at the linear zero of the SDF) and map_uv (dmtet.py:50-79: every tet owns one cell of an N x N atlas) in numpy; the
tet grid itself (data/tets/64_tets.npz, quartet output) is read from the reference checkout and is NOT copied: only the extracted
triangle meshes are written, as assets/dmtet64_<name>.npz (positions, triangles, per-face global index; the atlas is rebuilt from the
latter at load time, nvdiffrecmc_amd/scene.py).  /root/reference does not exist on the GPU box, hence the committed assets.

    python tools/make_dmtet_mesh.py [/root/reference] [--stats]

Variants (mesh_scale 2.4 as configs/nerf_lego.json:11):
  init   sdf = U[0,1) - 0.1 per vertex, seed 1234: the reference's own initialisation (dmtet.py:173) -- iteration 0 of every DMTet run
  mid    a union of seeded spheres blended with per-vertex noise: a surface with a rough skin, floaters and internal sheets, the
         state a DMTet run is in after a few hundred iterations
"""
import os
import sys

import numpy as np

# dmtet.py:20-40 -- triangle_table / num_triangles_table / base_tet_edges, as data
TRIANGLE_TABLE = np.array([
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
    [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
    [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
    [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]], dtype=np.int64)
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], dtype=np.int64)
BASE_TET_EDGES = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], dtype=np.int64)
TILED = {'mid': 2}        # variants extracted from the 2 x 2 x 2 tiling (a 128^3-class grid)


def marching_tets(pos, sdf, tets):
    """dmtet.py:91-141 in numpy: returns (verts [V,3] f32, faces [F,3] i64, face_gidx [F] i64)."""
    occ = sdf > 0
    occ4 = occ[tets]
    occ_sum = occ4.sum(-1)
    valid = (occ_sum > 0) & (occ_sum < 4)
    edges = tets[valid][:, BASE_TET_EDGES].reshape(-1, 2)
    edges = np.sort(edges, axis=1)                                  # sort_edges: (smaller, larger)
    uniq, inv = np.unique(edges, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    crossing = occ[uniq].sum(-1) == 1
    mapping = np.full(uniq.shape[0], -1, dtype=np.int64)
    mapping[crossing] = np.arange(int(crossing.sum()))
    idx_map = mapping[inv].reshape(-1, 6)
    ev = uniq[crossing]
    p = pos[ev].astype(np.float32)                                   # [E,2,3]
    s = sdf[ev].astype(np.float32).reshape(-1, 2, 1).copy()
    s[:, 1] *= -1.0
    denom = s.sum(1, keepdims=True)
    w = s[:, ::-1] / denom                                          # torch.flip(..., [1]) / denominator
    verts = (p * w).sum(1).astype(np.float32)
    tetindex = (occ4[valid] * (1 << np.arange(4))).sum(-1)
    nt = NUM_TRIANGLES[tetindex]
    one, two = nt == 1, nt == 2
    f1 = np.take_along_axis(idx_map[one], TRIANGLE_TABLE[tetindex[one]][:, :3], axis=1).reshape(-1, 3)
    f2 = np.take_along_axis(idx_map[two], TRIANGLE_TABLE[tetindex[two]][:, :6], axis=1).reshape(-1, 3)
    faces = np.concatenate((f1, f2), 0)
    gidx = np.arange(tets.shape[0], dtype=np.int64)[valid]
    face_gidx = np.concatenate((gidx[one] * 2, np.stack((gidx[two] * 2, gidx[two] * 2 + 1), -1).reshape(-1)), 0)
    return verts, faces, face_gidx


def map_uv(face_gidx, max_idx):
    """dmtet.py:50-79: (uvs [4 N^2, 2] f32, uv_idx [F,3] i64) -- every tet owns one cell of an N x N atlas, its (up to) two triangles
    the two halves of that cell."""
    N = int(np.ceil(np.sqrt((max_idx + 1) // 2)))
    import torch
    lin = torch.linspace(0, 1 - (1 / N), N, dtype=torch.float32).numpy()      # (torch's float32 linspace, as the reference: numpy's rounds differently)
    tex_y, tex_x = np.meshgrid(lin, lin, indexing='ij')
    pad = np.float32(0.9 / N)
    uvs = np.stack([tex_x, tex_y, tex_x + pad, tex_y, tex_x + pad, tex_y + pad, tex_x, tex_y + pad], -1).reshape(-1, 2).astype(np.float32)
    tet_idx = face_gidx // 2
    tri_idx = face_gidx % 2
    uv_idx = np.stack((tet_idx * 4, tet_idx * 4 + tri_idx + 1, tet_idx * 4 + tri_idx + 2), -1)
    return uvs, uv_idx


def sdf_variant(name, pos, rng):
    if name == 'init':
        return rng.random(pos.shape[0], dtype=np.float32) - np.float32(0.1)          # dmtet.py:173
    if name == 'mid':
        # union of 14 spheres (centres within 0.55 of the origin, radii 0.18-0.42) -> inside positive, like the reference's convention
        c = (rng.random((14, 3)) - 0.5) * 1.1
        r = 0.18 + 0.24 * rng.random(14)
        d = np.linalg.norm(pos[:, None, :] - c[None], axis=-1) - r[None]
        shape = -d.min(1)                                             # > 0 inside
        noise = rng.random(pos.shape[0]) - 0.5
        # a rough skin (noise of the order of one grid cell near the surface) + floaters / voids (sparse strong outliers)
        outl = (rng.random(pos.shape[0]) < 0.04) * (rng.random(pos.shape[0]) - 0.5) * 2.0
        return (shape + 0.06 * noise + outl * (np.abs(shape) < 0.5)).astype(np.float32)
    raise ValueError(name)


def tile_grid(verts, tets, k):
    """k x k x k translated half-... 1/k-scale copies of the unit tet grid, welded: the quartet grid of the reference is periodic (the
    vertices AND edges on opposite faces of the cube coincide under translation: checked below), so the result is a consistent tet
    grid of k times the resolution -- 2 gives the 128^3-class grid configs/nerf_lego.json:10 asks for (the 128 file is a download)."""
    q = np.round(verts.astype(np.float64) * 64).astype(np.int64)          # coordinates are multiples of 1/64 in [-1/2, 1/2]
    assert np.abs(q / 64.0 - verts).max() < 1e-6
    allq, allt = [], []
    for i in range(k):
        for j in range(k):
            for l in range(k):
                allq.append(q + 32 + 64 * np.array([i, j, l]))            # 0 .. 64 k
                allt.append(tets + len(allq[:-1]) * verts.shape[0])
    allq, allt = np.concatenate(allq), np.concatenate(allt)
    key = (allq[:, 0] * (64 * k + 1) + allq[:, 1]) * (64 * k + 1) + allq[:, 2]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    pos = (allq[first].astype(np.float64) / (64.0 * k) - 0.5).astype(np.float32)
    return pos, inv.reshape(-1)[allt]


def build(name, ref, scale=2.4, seed=1234):
    d = np.load(os.path.join(ref, 'data/tets/64_tets.npz'))
    verts, tets = d['vertices'].astype(np.float32), d['indices'].astype(np.int64)
    if name in TILED:
        verts, tets = tile_grid(verts, tets, TILED[name])
    pos = verts * np.float32(scale)                                 # dmtet.py:164
    rng = np.random.default_rng(seed)
    sdf = sdf_variant(name, pos, rng)
    v, f, gidx = marching_tets(pos, sdf, tets)
    # drop exactly degenerate faces (two corners on one grid edge cannot happen; zero-area ones can when the SDF is 0 at a vertex -- it never is here)
    return v, f, gidx, tets.shape[0]


def stats(v, f):
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=-1)
    e = np.stack((np.linalg.norm(b - a, axis=-1), np.linalg.norm(c - b, axis=-1), np.linalg.norm(a - c, axis=-1)), -1)
    aspect = e.max(-1) ** 2 / np.maximum(area, 1e-20)               # 2.31 for an equilateral triangle
    return {'verts': int(v.shape[0]), 'tris': int(f.shape[0]), 'area_min': float(area.min()), 'area_median': float(np.median(area)),
            'area_max': float(area.max()), 'aspect_median': float(np.median(aspect)), 'aspect_p99': float(np.percentile(aspect, 99)),
            'slivers_aspect_over_50': float((aspect > 50).mean())}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    ref = args[0] if args else '/root/reference'
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'assets')
    for name in ('init', 'mid'):
        v, f, gidx, n_tets = build(name, ref)
        st = stats(v, f)
        print(name, st)
        if '--stats' in sys.argv:
            continue
        np.savez_compressed(os.path.join(out, 'dmtet64_%s.npz' % name), v_pos=v, t_pos_idx=f.astype(np.int32), face_gidx=gidx.astype(np.int32),
                            n_tets=np.int64(n_tets))
        print('  ->', os.path.getsize(os.path.join(out, 'dmtet64_%s.npz' % name)), 'bytes')


if __name__ == '__main__':
    main()
