import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import optixutils as ou, scene as sc
dev = torch.device('cuda:0')
m = sc.load_mesh(os.environ.get('PROBE_MESH', 'bob'))
ctx = ou.OptiXContext()
ou.optix_build_bvh(ctx, m['v_pos'].to(dev), m['t_pos_idx'].to(dev), 1)
R = 1 << 22
g = torch.Generator().manual_seed(1)
ro = (torch.randn(R, 3, generator=g) * 0.25).to(dev)
rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
vis, cnt = ou.trace_visibility(ctx, ro, rd, count=True)
print('box/ray', cnt[0].item() / R, 'tri/ray', cnt[1].item() / R, 'visible', vis.float().mean().item(), ctx.bvh_info())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ou.trace_visibility(ctx, ro, rd)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('trace %.3f ms %.2f Grays/s' % (dt * 1e3, R / dt / 1e9))
