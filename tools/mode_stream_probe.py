"""Slow-mode survey, stream edition: in ONE fresh process, time the one-view traversal launch on torch's default stream and on
two newly created HIP streams (other hardware queues).  If a process that is slow on its default queue is fast on a new one, the
harness can fence the mode off by moving to a fresh stream.  usage: mode_stream_probe.py  (prints one line)"""
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
ctx.cache_visibility = False
torch.cuda.synchronize()


def run():
    ctx.set_profiling(True)
    for it in range(8):
        if it == 2:
            ctx.set_profiling(True)
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
    torch.cuda.current_stream().synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    ctx.set_profiling(False)
    return t


out = ['default %.3f' % run()]
for k in range(2):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out.append('new-stream-%d %.3f' % (k, run()))
    torch.cuda.synchronize()
out.append('default-again %.3f' % run())
print('traversal ms (%d triangles): %s' % (st.mesh['t_pos_idx'].shape[0], ' | '.join(out)))
