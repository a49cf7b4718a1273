"""What is slow in a slow process: the traversal KERNEL or the hipMemsetAsync that resets its chunk counters?  One process, the
one-view traversal launched for PROBE_SECONDS, cycling the three ways of resetting the counters from launch to launch:
  kernel  a 64-thread kernel inside the stage-2 timing bracket (the default since session M)
  inside  hipMemsetAsync of the 32 KB counter block inside the bracket (rounds 1-2)
  before  the same hipMemsetAsync BEFORE the bracket (it is then timed with stage 1, sample generation)
Prints, per way, the median / max of the stage-1 and stage-2 times."""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

res = int(os.environ.get('PROBE_RES', '800'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '3'))
nviews = int(os.environ.get('PROBE_VIEWS', '1'))
st = DirectLightingStep('bob', res, 8, view=list(range(nviews)), n_views=8, device='cuda:0', subdiv=subdiv)
with torch.no_grad():
    m = st.mask[..., None]
    kd = (st.kd_tex[st.texel].view(st.nv, res, res, 3) * m).contiguous()
    ks = (st.ks.view(1, 1, 1, 3) * m).contiguous()
    nrm = ru.prepare_shading_normal(st.gb_pos, st.view_pos, None, st.gb_smooth_nrm, st.gb_tangent, st.gb_geom_nrm)
    ro = st.gb_pos + nrm * 0.001
L = st.light
ctx = st.ctx
ctx.set_trace_variant(0)
cal = ctx.trace_selection()['ms']
modes = [('kernel', 0), ('inside', 2), ('before', 4)]
log = {name: [] for name, _ in modes}
t0 = time.perf_counter()
it = 0
while time.perf_counter() - t0 < float(os.environ.get('PROBE_SECONDS', '6')):
    name, flags = modes[it % 3]
    it += 1
    ctx.set_trace_flags(flags)
    ctx.set_profiling(True)
    for k in range(2):
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=k, shadow_scale=1.0)
    torch.cuda.synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    ctx.set_profiling(False)
    log[name].append((g, t))
ctx.set_trace_flags(0)
out = []
for name, _ in modes:
    g = [a for a, b in log[name]]
    t = [b for a, b in log[name]]
    out.append('%s: stage1 %.2f/%.2f stage2 %.2f/%.2f (n=%d)' % (name, statistics.median(g), max(g), statistics.median(t), max(t), len(t)))
print('first-launch calibration ms %s | median/max ms -- %s' % (' '.join('%.1f' % v if v is not None else 'n/a' for v in cal), ' | '.join(out)))
