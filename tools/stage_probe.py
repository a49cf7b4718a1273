"""Per-stage HIP-event times of the env-shade op (gen / trace / shade, forward and backward) on the benchmark workload,
for a list of NVDR_PBLOCKS / NVDR_DEBUG settings given as argv: e.g.  stage_probe.py 4,4,4 6,4,5 'dbg=8'"""
import os as _os; _os.environ.setdefault('NVDR_TUNING', '1')
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru
from tools.gpu_tenancy import cpu_state
import time as _time

n = int(os.environ.get('PROBE_N', '8'))
res = int(os.environ.get('PROBE_RES', '512'))
mesh = os.environ.get('PROBE_MESH', 'bob')
subdiv = int(os.environ.get('PROBE_SUBDIV', '0'))
nviews = int(os.environ.get('PROBE_VIEWS', '1'))
st = DirectLightingStep(mesh, res, n, view=list(range(nviews)), n_views=8, device='cuda:0', subdiv=subdiv, retrace_backward=True)
m = st.mask[..., None]
with torch.no_grad():
    _, ro, _, nrm, _, kd, ks = st.shade_inputs()
L = st.light
ou.ops._optix_env_shade_func.cache_visibility = False

def run(iters=12):
    st.ctx.set_profiling(True)
    for it in range(iters):
        if it == 2: st.ctx.set_profiling(True)
        if os.environ.get('PROBE_REBUILD'): ou.optix_build_bvh(st.ctx, st.mesh['v_pos'], st.mesh['t_pos_idx'], rebuild=1)   # as the training iteration does
        g = [t.clone().requires_grad_(True) for t in (st.gb_pos, nrm, kd, ks, L.base.detach())]
        d, s = ou.optix_env_shade(st.ctx, st.mask, ro, g[0], g[1], st.view_pos, g[2], g[3], g[4], L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=n, rnd_seed=it, shadow_scale=1.0)
        torch.autograd.backward([d, s], [torch.ones_like(d), torch.ones_like(s)])
    torch.cuda.synchronize()
    nf, f = st.ctx.stage_times(backward=False)
    nb, b = st.ctx.stage_times(backward=True)
    st.ctx.set_profiling(False)
    return f, b

print('workload: %s %dx%d n=%d covered=%d triangles=%d | NVDR_TRACE_XCD=%s' % (mesh, res, res, n, st.covered, st.mesh['t_pos_idx'].shape[0], os.environ.get('NVDR_TRACE_XCD', '-')))
# BVH rebuild time (stream-ordered, no host sync inside)
for _ in range(3): ou.optix_build_bvh(st.ctx, st.mesh['v_pos'], st.mesh['t_pos_idx'], rebuild=1)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ou.optix_build_bvh(st.ctx, st.mesh['v_pos'], st.mesh['t_pos_idx'], rebuild=1)
e1.record(); torch.cuda.synchronize()
print('bvh rebuild %.3f ms' % (e0.elapsed_time(e1) / 10))
for cfg in sys.argv[1:] or ['4,4,4']:
    os.environ.pop('NVDR_PBLOCKS', None); os.environ['NVDR_DEBUG'] = '0'
    for tok in cfg.split(';'):
        if tok.startswith('dbg='): os.environ['NVDR_DEBUG'] = tok[4:]
        elif tok: os.environ['NVDR_PBLOCKS'] = tok
    c0, w0 = cpu_state(), _time.perf_counter()
    f, b = run()
    c1, w1 = cpu_state(), _time.perf_counter()
    print('%-16s fwd gen %.3f trace %.3f shade %.3f | bwd gen %.3f trace %.3f shade %.3f  (ms)' % ((cfg,) + tuple(f) + tuple(b)))
    print('host side: %.1f ms wall per fwd+bwd pass (12 passes), torch threads %d, cpu state before %s after %s' % ((w1 - w0) / 12 * 1e3, torch.get_num_threads(), c0, c1))
os.environ.pop('NVDR_PBLOCKS', None); os.environ['NVDR_DEBUG'] = '0'
st.ctx.set_profiling(True)               # the bracket of the COUNTING launch too (stage 2 = counting kernel + canonical binary walk)
P, nb, nt, nr = ou.ops.env_shade_traversal_counts(st.ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                                                  n_samples_x=n, rnd_seed=0)
_n, _cf = st.ctx.stage_times(backward=False)
st.ctx.set_profiling(False)
print('counting launch, event brackets: gen %.3f trace+binary-walk %.3f shade %.3f ms' % tuple(_cf))
tot, mx, nw = ou.ops.env_shade_traversal_counts.balance
print('counting build: %d rays traversed of %d, %.1f box %.2f tri tests/ray; per-wave busy time mean %.1f us max %.1f us over %d waves'
      % (nr, 2 * n * n * P, nb / nr, nt / nr, tot / max(nw, 1) / 100.0, mx / 100.0, nw))
_lb = ou.ops.env_shade_traversal_counts.leaf_batches
print('production walk: %.2f node steps per ray; %d triangle-test batches, %.1f of 64 lanes filled on average'
      % (ou.ops.env_shade_traversal_counts.node_steps / max(nr, 1), _lb[0], _lb[1] / max(_lb[0], 1)))
wt = ou.ops.env_shade_traversal_counts.wave_ticks.double()
t0 = wt[:, 0].min()
b, e = (wt[:, 0] - t0) / 100.0, (wt[:, 1] - t0) / 100.0
q = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], dtype=torch.float64)
print('shader clock during the counting launch: %.0f MHz, XCD mask 0x%x' % (ou.ops.env_shade_traversal_counts.clock_mhz, ou.ops.env_shade_traversal_counts.xcd_mask))
print('wave begin (us) quantiles', [round(x, 1) for x in torch.quantile(b, q).tolist()])
print('wave end   (us) quantiles', [round(x, 1) for x in torch.quantile(e, q).tolist()])
print('wave busy  (us) quantiles', [round(x, 1) for x in torch.quantile(e - b, q).tolist()])
