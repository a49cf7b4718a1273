"""Slow-mode survey, kernel-copy edition: one fresh process, the one-view traversal launch timed with each of the three identical
copies of env_trace_kernel (nvdr_ctx_set_trace_variant), plus the program counters they ran at."""
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


def run():
    ctx.set_profiling(True)
    for it in range(6):
        if it == 2:
            ctx.set_profiling(True)
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
    torch.cuda.synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    ctx.set_profiling(False)
    return t


out = []
for k in (0, 1, 2, 0):
    ctx.set_trace_variant(k)
    out.append('copy%d %.3f' % (k, run()))
ou.ops.env_shade_traversal_counts(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=8, rnd_seed=0)
pcs = ctx.trace_pcs()
print('traversal ms: %s | pc %s' % (' | '.join(out), ' '.join('%x' % p for p in pcs)))
