"""G-buffer kernel alone: HIP-event time of nvdr_render_gbuffer for one view and eight views (NVDR_GB_MODE: bit 0 tiles, bit 1 claimed units)."""
import os, sys
os.environ.setdefault('NVDR_TUNING', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvdiffrecmc_amd import optixutils as ou, scene as sc
dev = torch.device('cuda:0')
mesh_name, subdiv, res = os.environ.get('PROBE_MESH', 'bob'), int(os.environ.get('PROBE_SUBDIV', '3')), int(os.environ.get('PROBE_RES', '800'))
m = sc.load_mesh(mesh_name)
if subdiv:
    m = sc.subdivide_mesh(m, subdiv)
md = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in m.items()}
ctx = ou.OptiXContext()
ou.optix_build_bvh(ctx, md['v_pos'], md['t_pos_idx'], 1)
for nv in (1, 8):
    cams = [sc.camera(v, 8) for v in range(nv)]
    mvp = torch.stack([c[1] for c in cams]).to(dev)
    cam = torch.stack([sc.camera_rays(c[0]) for c in cams]).to(dev)
    for _ in range(3):
        gb = ou.render_gbuffer(ctx, md, mvp, cam, (res, res))
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); gb = ou.render_gbuffer(ctx, md, mvp, cam, (res, res)); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print('NVDR_GB_MODE=%s %s subdiv %d %dx%d, %d view(s): median %.3f ms (%.3f per view), covered %d' % (os.environ.get('NVDR_GB_MODE', 'default'), mesh_name, subdiv, res, res, nv, ms[10], ms[10] / nv, int((gb['rast'][..., 3] > 0).sum())))
