"""Does the slow state of the traversal kernel re-roll when the GPU has been idle?  One process runs BURSTS of forward env-shade
launches separated by idle gaps of varying length and prints, per burst: the gap before it, the median stage-1 and stage-2 times,
then (one counting launch) the shader clock and the mean busy time of the wavefronts per XCD."""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru
from tools.gpu_tenancy import snapshot

res = int(os.environ.get('PROBE_RES', '512'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '0'))
nviews = int(os.environ.get('PROBE_VIEWS', '1'))
burst = int(os.environ.get('PROBE_BURST', '30'))
gaps = [float(v) for v in os.environ.get('PROBE_GAPS', '0,0.05,0.2,0.5,1,2').split(',')]
n_bursts = int(os.environ.get('PROBE_BURSTS', '18'))
st = DirectLightingStep('bob', res, 8, view=list(range(nviews)), n_views=8, device='cuda:0', subdiv=subdiv, retrace_backward=True)
with torch.no_grad():
    m = st.mask[..., None]
    _, ro, _, nrm, _, kd, ks = st.shade_inputs()
L = st.light
ctx = st.ctx
f = ou.ops.env_shade_traversal_counts
print('tenancy before: ' + snapshot())
for b in range(n_bursts):
    gap = gaps[b % len(gaps)]
    torch.cuda.synchronize()
    time.sleep(gap)
    ts = []
    for k in range(burst):
        ctx.set_profiling(True)
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=k, shadow_scale=1.0)
        if k % 10 == 9 or k == burst - 1:
            torch.cuda.synchronize()
        if k % 10 == 9 or k == burst - 1:
            n, (g, t, sh) = ctx.stage_times(backward=False)
            ts.append((g, t))
            ctx.set_profiling(False)
    f(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols, n_samples_x=8, rnd_seed=0)
    wt, xcd = f.wave_ticks.double(), f.wave_xcd
    busy = (wt[:, 1] - wt[:, 0]) / 100.0
    span = (wt[:, 1].max() - wt[:, 0].min()).item() / 100.0
    per = ' '.join('%.0f' % busy[xcd == x].mean().item() for x in range(8) if (xcd == x).any())
    print('burst %2d after %.2f s idle: stage1 %.3f stage2 %.3f ms | counting launch %.0f us at %.0f MHz, wave busy us per XCD: %s'
          % (b, gap, statistics.median(a for a, _ in ts), statistics.median(t for _, t in ts), span, f.clock_mhz, per))
    if statistics.median(t for _, t in ts) > float(os.environ.get('PROBE_SLOW_MS', '1e9')):
        print('   slow burst, tenancy: ' + snapshot())
print('tenancy after: ' + snapshot())
