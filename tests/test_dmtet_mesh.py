"""The DMTet-shaped benchmark meshes: the numpy restatement of marching_tets / map_uv that extracts them
(tools/make_dmtet_mesh.py, nvdiffrecmc_amd/scene.py:dmtet_atlas) against vectors of the reference's own functions
(tests/golden/dmtet_reference.npz, written by tools/make_golden.py gen_dmtet from geometry/dmtet.py:50-141), and the committed assets."""
import hashlib
import os

import numpy as np
import pytest
import torch

from nvdiffrecmc_amd import scene as sc
from tools import make_dmtet_mesh as mk

REF_GRID = '/root/reference/data/tets/64_tets.npz'


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('name', ['random', 'sphere'])
def test_marching_tets_restatement_vs_reference_vectors(name, golden_dir):
    g = np.load(os.path.join(golden_dir, 'dmtet_reference.npz'))
    verts, tets, sdf = g['grid/vertices'], g['grid/indices'], g[name + '/sdf']
    v, f, gidx = mk.marching_tets(verts * np.float32(2.4), sdf, tets)
    assert v.shape == g[name + '/verts'].shape and f.shape == g[name + '/faces'].shape
    assert np.array_equal(f.astype(np.int32), g[name + '/faces'])                 # same vertex numbering, same triangle order
    assert np.array_equal(v, g[name + '/verts'])                                   # the same float32 arithmetic: bit for bit
    uvs, uv_idx = mk.map_uv(gidx, tets.shape[0] * 2)
    assert np.array_equal(uv_idx.astype(np.int32), g[name + '/uv_idx'])
    assert _sha(uvs) == str(g[name + '/uvs_sha256']) and np.array_equal(uvs[:64], g[name + '/uvs_head'])
    # the loader's atlas (rebuilt from the per-face global index the asset stores) is the same one
    v_tex, t_tex = sc.dmtet_atlas(gidx, tets.shape[0])
    assert np.array_equal(v_tex.numpy(), uvs) and np.array_equal(t_tex.numpy(), uv_idx.astype(np.int32))


@pytest.mark.parametrize('name,tris', [('dmtet64_init', 77705), ('dmtet64_mid', 140114)])
def test_committed_dmtet_assets(name, tris):
    m = sc.load_mesh(name)
    T, V = m['t_pos_idx'].shape[0], m['v_pos'].shape[0]
    assert T == tris and m['t_tex_idx'].shape[0] == T
    t = m['t_pos_idx'].long()
    assert int(t.min()) == 0 and int(t.max()) == V - 1 and torch.unique(t).numel() == V       # every vertex is used
    assert int(m['t_tex_idx'].max()) < m['v_tex'].shape[0] and float(m['v_tex'].min()) >= 0.0 and float(m['v_tex'].max()) <= 1.0
    for k in ('v_pos', 'v_nrm', 'v_tng'):
        assert torch.isfinite(m[k]).all()
    assert float(m['v_pos'].abs().max()) <= 1.2 + 1e-6                           # mesh_scale 2.4 (configs/nerf_lego.json:11)
    st = mk.stats(m['v_pos'].numpy(), t.numpy())
    # what makes these meshes unlike the subdivided stand-ins: a wide spread of areas and a tail of slivers
    assert st['area_max'] / st['area_median'] > 4.0 and st['area_min'] < 1e-3 * st['area_median'] and st['slivers_aspect_over_50'] > 0.01


@pytest.mark.skipif(not os.path.exists(REF_GRID), reason='needs the reference checkout (the tet grid is not copied into this repository)')
@pytest.mark.parametrize('name', ['init', 'mid'])
def test_assets_are_what_the_tool_extracts_from_the_reference_grid(name):
    v, f, gidx, n_tets = mk.build(name, '/root/reference')
    d = np.load(os.path.join(os.path.dirname(sc.__file__), '..', 'assets', 'dmtet64_%s.npz' % name))
    assert np.array_equal(d['v_pos'], v) and np.array_equal(d['t_pos_idx'], f.astype(np.int32))
    assert np.array_equal(d['face_gidx'], gidx.astype(np.int32)) and int(d['n_tets']) == n_tets


@pytest.mark.skipif(not os.path.exists(REF_GRID), reason='needs the reference checkout')
def test_the_reference_grid_is_periodic_so_its_tiling_is_a_consistent_grid():
    """tile_grid welds translated copies: legal because the vertices AND the edges on opposite faces of the quartet cube coincide."""
    d = np.load(REF_GRID)
    q = np.round(d['vertices'].astype(np.float64) * 64).astype(np.int64)
    t = d['indices']
    E = np.unique(np.sort(t[:, [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3]].reshape(-1, 2), 1), axis=0)
    for ax in range(3):
        oth = [a for a in range(3) if a != ax]
        sets = []
        for val in (-32, 32):
            on = q[:, ax] == val
            e = E[on[E[:, 0]] & on[E[:, 1]]]
            a, b = q[e[:, 0]][:, oth], q[e[:, 1]][:, oth]
            sets.append(({tuple(p) for p in q[on][:, oth]},
                         {tuple(sorted((tuple(x), tuple(y)))) for x, y in zip(a.tolist(), b.tolist())}))
        assert sets[0] == sets[1] and len(sets[0][0]) > 1000
    pos, tets = mk.tile_grid(d['vertices'].astype(np.float32), t.astype(np.int64), 2)
    assert tets.shape[0] == 8 * t.shape[0] and pos.shape[0] < 8 * q.shape[0]
    # every tet of the tiling keeps a positive volume of 1/8 of its source tet
    vol = lambda p, tt: np.einsum('ij,ij->i', np.cross(p[tt[:, 1]] - p[tt[:, 0]], p[tt[:, 2]] - p[tt[:, 0]]), p[tt[:, 3]] - p[tt[:, 0]])
    v0, v1 = vol(d['vertices'].astype(np.float64), t), vol(pos.astype(np.float64), tets)
    assert np.allclose(np.tile(v0, 8) / 8.0, v1, rtol=1e-5)
