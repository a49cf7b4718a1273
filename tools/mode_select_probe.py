"""Slow-mode survey, instance-selection edition: one fresh process renders one forward pass (the library times the five
instances of the traversal kernel on that launch and selects one: OptiXContext.trace_selection), then every instance is
timed explicitly over a few launches, then the selected one with the per-XCD chunk dealing.  One line per process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

res = int(os.environ.get('PROBE_RES', '512'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '0'))
nviews = int(os.environ.get('PROBE_VIEWS', '1'))
st = DirectLightingStep('bob', res, 8, view=list(range(nviews)), n_views=8, device='cuda:0', subdiv=subdiv)   # renders the target: first forward launch
sel = st.ctx.trace_selection()
with torch.no_grad():
    m = st.mask[..., None]
    kd = (st.kd_tex[st.texel].view(st.nv, res, res, 3) * m).contiguous()
    ks = (st.ks.view(1, 1, 1, 3) * m).contiguous()
    nrm = ru.prepare_shading_normal(st.gb_pos, st.view_pos, None, st.gb_smooth_nrm, st.gb_tangent, st.gb_geom_nrm)
    ro = st.gb_pos + nrm * 0.001
L = st.light
ctx = st.ctx


def run(iters=4):
    ctx.set_profiling(True)
    for it in range(iters):
        if it == 1:
            ctx.set_profiling(True)          # drop the first launch
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
    torch.cuda.synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    ctx.set_profiling(False)
    return t


fmt = lambda v: 'n/a' if v is None else '%.3f' % v
out = []
for k in range(5):
    try:
        ctx.set_trace_variant(k)
        out.append('%.3f' % run())
    except RuntimeError as e:
        out.append('n/a')
ctx.set_trace_variant(-1)
t_sel = run()
ctx.set_trace_xcd_partition(True)
t_xcd = run()
ctx.set_trace_xcd_partition(False)
pcs = ctx.trace_pcs()
print('select: choice %d decided %d attempts %d calib ms [%s] | explicit ms [%s] | selected %.3f xcd-partitioned %.3f | pc %s'
      % (sel['choice'], sel['decided'], sel['attempts'], ' '.join(fmt(v) for v in sel['ms']), ' '.join(out), t_sel, t_xcd,
         ' '.join('%x' % p for p in pcs)))
