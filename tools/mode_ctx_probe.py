"""Slow-mode survey INSIDE one process: does the 1.7x slow mode of the one-view traversal launch belong to the process or to
the context (i.e. to where its buffers landed)?  Creates K contexts one after the other on the same scene and prints the
traversal stage time of each.  usage: mode_ctx_probe.py [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
st = DirectLightingStep('bob', 512, 8, view=[0], n_views=8, device='cuda:0')
with torch.no_grad():
    m = st.mask[..., None]
    kd = (st.kd_tex[st.texel].view(st.nv, 512, 512, 3) * m).contiguous()
    ks = (st.ks.view(1, 1, 1, 3) * m).contiguous()
    nrm = ru.prepare_shading_normal(st.gb_pos, st.view_pos, None, st.gb_smooth_nrm, st.gb_tangent, st.gb_geom_nrm)
    ro = st.gb_pos + nrm * 0.001
L = st.light
out = []
keep = []
for k in range(K):
    ctx = st.ctx if k == 0 else ou.OptiXContext()
    if k:
        if k % 2 == 0:
            keep.append(torch.empty((37 + 11 * k) << 20, dtype=torch.uint8, device='cuda'))   # shift where the next buffers land
        ou.optix_build_bvh(ctx, st.mesh['v_pos'], st.mesh['t_pos_idx'], 1)
    ctx.cache_visibility = False
    ctx.set_profiling(True)
    for it in range(8):
        if it == 2:
            ctx.set_profiling(True)
        d, s = ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
    torch.cuda.synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    out.append('%.3f' % t)
print('one process, %d contexts: traversal ms per context: %s' % (K, ' '.join(out)))
