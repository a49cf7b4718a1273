"""Slow-mode survey, XCD edition: one fresh process, the one-view traversal launch timed normally, then the counting build's
per-wavefront busy times grouped by the XCD the wavefront ran on.  Are all XCDs slow in a slow process, or one?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

res = int(os.environ.get('PROBE_RES', '512'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '0'))
st = DirectLightingStep('bob', res, 8, view=[0], n_views=8, device='cuda:0', subdiv=subdiv)
with torch.no_grad():
    m = st.mask[..., None]
    kd = (st.kd_tex[st.texel].view(st.nv, res, res, 3) * m).contiguous()
    ks = (st.ks.view(1, 1, 1, 3) * m).contiguous()
    nrm = ru.prepare_shading_normal(st.gb_pos, st.view_pos, None, st.gb_smooth_nrm, st.gb_tangent, st.gb_geom_nrm)
    ro = st.gb_pos + nrm * 0.001
L = st.light
ctx = st.ctx
ctx.set_profiling(True)
for it in range(6):
    if it == 2:
        ctx.set_profiling(True)
    ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                       n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
torch.cuda.synchronize()
n, (g, t, sh) = ctx.stage_times(backward=False)
ctx.set_profiling(False)
ou.ops.env_shade_traversal_counts(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=8, rnd_seed=0)
f = ou.ops.env_shade_traversal_counts
wt, xcd = f.wave_ticks.double(), f.wave_xcd
busy = (wt[:, 1] - wt[:, 0]) / 100.0          # microseconds
per = ['%d:%.0f/%.0f' % (x, busy[xcd == x].mean().item(), busy[xcd == x].max().item()) for x in range(8) if (xcd == x).any()]
print('trace %.3f ms | counting launch: clock %.0f MHz, wave busy us per XCD (mean/max): %s' % (t, f.clock_mhz, ' '.join(per)))
