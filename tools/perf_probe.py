"""Kernel-level timing probe of the env-shade op on the benchmark workload (bob 512^2, n=8) with the NVDR_DEBUG knobs."""
import os as _os; _os.environ.setdefault('NVDR_TUNING', '1')
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

n = int(os.environ.get('PROBE_N', '8'))
res = int(os.environ.get('PROBE_RES', '512'))
mesh = os.environ.get('PROBE_MESH', 'bob')
st = DirectLightingStep(mesh, res, n, view=0, n_views=8, device='cuda:0', retrace_backward=True)
m = st.mask[..., None]
with torch.no_grad():
    _, ro, _, nrm, _, kd, ks = st.shade_inputs()
L = st.light
rays = st.rays_per_pass()

def run(backward, cache, iters=10):
    ou.ops._optix_env_shade_func.cache_visibility = cache
    ts_f, ts_b = [], []
    for it in range(iters + 2):
        g = [t.clone().requires_grad_(True) for t in (st.gb_pos, nrm, kd, ks, L.base.detach())]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        d, s = ou.optix_env_shade(st.ctx, st.mask, ro, g[0], g[1], st.view_pos, g[2], g[3], g[4], L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=n, rnd_seed=it, shadow_scale=1.0)
        e[1].record()
        if backward:
            gd, gs = torch.ones_like(d), torch.ones_like(s)
            e[2].record()
            torch.autograd.backward([d, s], [gd, gs])
            e[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            ts_f.append(e[0].elapsed_time(e[1]))
            if backward: ts_b.append(e[2].elapsed_time(e[3]))
    f = sum(ts_f) / len(ts_f)
    b = sum(ts_b) / len(ts_b) if ts_b else float('nan')
    return f, b

print('workload: %s %dx%d n=%d covered=%d rays/pass=%d' % (mesh, res, res, n, st.covered, rays))
for name, dbg, cache in (('normal', '0', False), ('skip-trace', '1', False), ('skip-atomics', '2', False), ('skip-trace+atomics', '3', False), ('vis-cache', '0', True)):
    os.environ['NVDR_DEBUG'] = dbg
    f, b = run(True, cache)
    print('%-20s fwd %.3f ms (%.2f Grays/s)   bwd %.3f ms (%.2f Grays/s)' % (name, f, rays / f / 1e6, b, rays / b / 1e6))
os.environ['NVDR_DEBUG'] = '0'
