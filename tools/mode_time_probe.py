"""Is the slow mode of the traversal kernel a PHASE?  One process launches the one-view traversal continuously for ~14 s and logs the
time of every launch; at t = 4 s it starts a child process that allocates and touches PROBE_CHILD_GB of device memory and exits
(= what the previous process of a benchmark loop does while this one starts); at t = 10 s the process itself allocates and frees the
same amount with hipMalloc / hipFree.  Prints the launch times in half-second buckets: median / max ms."""
import os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

res = int(os.environ.get('PROBE_RES', '800'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '3'))
gb = float(os.environ.get('PROBE_CHILD_GB', '8'))
t_start = time.perf_counter()
st = DirectLightingStep('bob', res, 8, view=[0], n_views=8, device='cuda:0', subdiv=subdiv, retrace_backward=True)
with torch.no_grad():
    m = st.mask[..., None]
    _, ro, _, nrm, _, kd, ks = st.shade_inputs()
L = st.light
ctx = st.ctx
print('set-up %.1f s after process start' % (time.perf_counter() - t_start))
log = []           # (seconds since t0, trace ms)
events = []
t0 = time.perf_counter()
child = None
did_child = did_free = False
while True:
    now = time.perf_counter() - t0
    if now > float(os.environ.get('PROBE_SECONDS', '14')):
        break
    if not did_child and now > 4.0:
        did_child = True
        child = subprocess.Popen([sys.executable, '-c',
                                  'import torch,time; x=torch.empty(int(%f*2**30), dtype=torch.uint8, device="cuda"); x.fill_(1); '
                                  'torch.cuda.synchronize(); time.sleep(0.5)' % gb])
        events.append((now, 'child started (%g GB)' % gb))
    if child is not None and child.poll() is not None:
        events.append((now, 'child exited'))
        child = None
    if not did_free and now > 10.0:
        did_free = True
        x = torch.empty(int(gb * 2**30), dtype=torch.uint8, device='cuda')
        x.fill_(1)
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
        events.append((time.perf_counter() - t0, 'own %g GB allocated, touched and freed (hipFree)' % gb))
    ctx.set_profiling(True)
    for it in range(3):
        ou.optix_env_shade(ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                           n_samples_x=8, rnd_seed=it, shadow_scale=1.0)
    torch.cuda.synchronize()
    n, (g, t, sh) = ctx.stage_times(backward=False)
    ctx.set_profiling(False)
    log.append((now, t))
print('events: ' + '; '.join('%.1f s %s' % e for e in events))
import statistics
b = 0.0
line = []
while b < log[-1][0] + 0.5:
    v = [t for (s, t) in log if b <= s < b + 0.5]
    if v:
        line.append('%.1fs %.2f/%.2f' % (b, statistics.median(v), max(v)))
    b += 0.5
print('trace ms per 0.5 s bucket (median/max of 3-launch averages): ' + '  '.join(line))
