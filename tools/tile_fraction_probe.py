"""How sparse is the gradient of a nearest-texel texture, and at which granularity?  For the views of a preset: the share of the texels of a
[R,R,3] texture some covered pixel looks up, and the share of the TILES of t consecutive texels that hold such a texel, per view and for the
union of the batch (what a data-parallel exchange over one-view-per-GPU ranks would have to send).
    python tools/tile_fraction_probe.py [bob512|dmtet800|spot512x256|hotdog512x256]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nvdiffrecmc_amd import optixutils as ou, scene as sc

name = sys.argv[1] if len(sys.argv) > 1 else 'bob512'
pre = bench.PRESETS[name]
dev = torch.device('cuda:0')
mesh = sc.load_mesh(pre['mesh'], device='cpu')
if pre['subdiv']:
    mesh = sc.subdivide_mesh(mesh, pre['subdiv'], cache_key=pre['mesh'])
mesh = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in mesh.items()}
ctx = ou.OptiXContext()
ou.optix_build_bvh(ctx, mesh['v_pos'], mesh['t_pos_idx'], rebuild=1)
nv, res, R = pre['batch'], pre['res'], pre.get('tex_res', 1024)
cams = [sc.camera(v, nv) for v in range(nv)]
gb = ou.render_gbuffer(ctx, mesh, torch.stack([c[1] for c in cams]).to(dev), torch.stack([sc.camera_rays(c[0]) for c in cams]).to(dev), (res, res))
cov = gb['rast'][..., 3] > 0
tc = gb['gb_texc']
ix = (tc[..., 0] * R).long().clamp(0, R - 1)
iy = ((1.0 - tc[..., 1]) * R).long().clamp(0, R - 1)
texel = iy * R + ix
print('%s: %d views of %dx%d, texture %dx%d, covered pixels per view %s' % (name, nv, res, res, R, R, [int(c.sum()) for c in cov]))
print('%-28s %s' % ('tile (texels)', '  '.join('%7d' % t for t in (1, 4, 8, 16, 64, 256))))
def frac(mask_views, t):
    hit = torch.zeros(R * R // t, dtype=torch.bool, device=dev)
    for v in mask_views:
        hit[texel[v][cov[v]] // t] = True
    return float(hit.float().mean())
for label, views in [('view 0', [0]), ('view %d' % (nv // 2), [nv // 2]), ('union of all %d views' % nv, list(range(nv)))]:
    print('%-28s %s' % (label, '  '.join('%6.1f%%' % (100 * frac(views, t)) for t in (1, 4, 8, 16, 64, 256))))
# square tiles (8 x 8 texels) instead of runs of 64 in a row
def frac_sq(views, s):
    hit = torch.zeros((R // s) * (R // s), dtype=torch.bool, device=dev)
    for v in views:
        hit[(iy[v][cov[v]] // s) * (R // s) + ix[v][cov[v]] // s] = True
    return float(hit.float().mean())
print('%-28s %s' % ('square tiles s x s, union', '  '.join('%dx%d %5.1f%%' % (s, s, 100 * frac_sq(list(range(nv)), s)) for s in (2, 4, 8, 16))))
print('%-28s %s' % ('square tiles s x s, view 0', '  '.join('%dx%d %5.1f%%' % (s, s, 100 * frac_sq([0], s)) for s in (2, 4, 8, 16))))
