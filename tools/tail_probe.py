"""Where the traversal kernel's time goes, from its COUNTING build (csrc/trace_kernel.h, template <true>):

  * per-phase shader-clock cycles of the wavefront loop (refill / node fetch / box arithmetic / hit masks + stack / leaf-queue rounds /
    triangle batches), summed over the wavefronts -- VERDICT r5 item 3 asked for exactly this table;
  * the launch's timeline from the per-wavefront (begin, end) ticks: how long the wavefronts take to start, how long the last ones run
    after the first has run out of work (the tail a small launch cannot amortise), how busy the slots are over the kernel's span.

usage: PROBE_CASES="bob:512:0:1,bob:512:0:8,bob:800:3:1,dmtet64_mid:800:0:1" python tools/tail_probe.py [out.md]     (mesh:res:subdiv:views)
The phase-clock builds are NOT the production kernel: shares, not absolute times."""
import os
import sys

os.environ.setdefault('NVDR_TUNING', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nvdiffrecmc_amd.trainer import DirectLightingStep  # noqa: E402
from nvdiffrecmc_amd import optixutils as ou  # noqa: E402

cases = os.environ.get('PROBE_CASES', 'bob:512:0:1,bob:512:0:8,bob:800:3:1,dmtet64_mid:800:0:1').split(',')
n = int(os.environ.get('PROBE_N', '8'))
lines = ['# Traversal kernel: per-phase cycles and launch timeline (counting build)', '',
         '`python tools/tail_probe.py` -- the two phase-clock builds of `env_trace_kernel` (`env_trace_phase_kernel<1|2>`, csrc/trace_kernel.h: clock reads at wave-uniform points; '
         'build 2 also inside the node step, behind a wait for the node), run behind the counting kernel on the same rays.  They are not the production kernel (13 / 24 spilled dwords '
         'against 2, a scalar memory read per clock): the factor against the production launch is given per case.', '']
for case in cases:
    mesh, res, subdiv, views = case.split(':')
    res, subdiv, views = int(res), int(subdiv), int(views)
    st = DirectLightingStep(mesh, res, n, view=list(range(views)), n_views=8, device='cuda:0', subdiv=subdiv, retrace_backward=True)
    for _ in range(3):
        st.step(1)
    with torch.no_grad():
        _, ro, _, nrm, _, kd, ks = st.shade_inputs()
        L = st.light
        f = ou.ops.env_shade_traversal_counts
        P, n_box, n_tri, n_traced = f(st.ctx, st.mask, ro, st.gb_pos, nrm, st.view_pos, kd, ks, L.base, L._pdf, L.rows[:, 0], L.cols, n_samples_x=n, rnd_seed=0)
    ph, ps = f.phases, f.phases_split
    # the production kernel's own time (HIP events around the traversal stage of plain iterations) to put the phase builds' distortion into numbers
    st.ctx.set_profiling(True)
    for _ in range(6):
        st.step(1)
    torch.cuda.synchronize()
    _, (gen_ms, trace_ms, shade_ms) = st.ctx.stage_times(backward=False)
    st.ctx.set_profiling(False)
    waves = f.balance[2]
    wt = f.wave_ticks.double()
    t0, t1 = float(wt[:, 0].min()), float(wt[:, 1].max())
    span = t1 - t0
    begin = (wt[:, 0] - t0) / span
    end = (wt[:, 1] - t0) / span
    busy = float((wt[:, 1] - wt[:, 0]).sum()) / (span * wt.shape[0])
    q = lambda x, p: float(torch.quantile(x, p))
    iters, step_iters, total1, total2 = ph[4], ph[5], ph[6], ps[6]
    prod_cycles = trace_ms * 1e-3 * f.clock_mhz * 1e6 * busy * waves          # wave-cycles of the production launch, estimated: duration x clock x slots busy
    # who ends when: wavefront w serves dealing counter w % 64 (trace_kernel.h ChunkDealer); per counter and per XCD, the mean end of its wavefronts
    wid = torch.arange(wt.shape[0])
    per_counter = torch.stack([end[(wid % 64) == qq].mean() for qq in range(64)])
    xcd = f.wave_xcd[:wt.shape[0]]
    per_xcd = [float(end[xcd == k].mean()) if bool((xcd == k).any()) else float('nan') for k in range(8)]
    spread_within = float(torch.stack([end[(wid % 64) == qq].std() for qq in range(64)]).mean())
    who = ('mean end of the wavefronts of a dealing counter: min %.1f / median %.1f / max %.1f %% of the span over the 64 counters (spread inside a counter: %.1f %% std); per XCD: %s'
           % (100 * float(per_counter.min()), 100 * float(per_counter.median()), 100 * float(per_counter.max()), 100 * spread_within,
              ' '.join('%.1f' % (100 * v) for v in per_xcd)))
    lines += ['## %s, %dx%d, %d view(s): %d triangles, %d rays, %d wavefronts (%.0f rays each), %.2f node steps per ray'
              % (mesh + (' subdivided %dx' % subdiv if subdiv else ''), res, res, views, int(st.mesh['t_pos_idx'].shape[0]), n_traced, waves, n_traced / max(waves, 1),
                 f.node_steps / max(n_traced, 1)), '',
              'production kernel %.3f ms per launch (HIP events); wave-cycles: production ~%.3g (duration x %.0f MHz x busy slots), phase build 1 %.3g (x %.2f), phase build 2 %.3g (x %.2f)'
              % (trace_ms, prod_cycles, f.clock_mhz, total1, total1 / prod_cycles, total2, total2 / prod_cycles), '',
              '| phase | build 1: cycles | share | per loop iteration | build 2 (node step split; waits for the node): cycles | share |', '|---|---|---|---|---|---|']
    rows1 = [('refill (votes, chunk claims, flush decision, ray fetch + set-up)', ph[0], ps[0]), ('node step: pop, address, node fetch', None, ps[1]),
             ('node step: box arithmetic of the eight children', None, ps[2]), ('node step: hit masks, push, group bookkeeping', None, ps[3]),
             ('node step, whole', ph[1], ps[1] + ps[2] + ps[3]), ('leaf-queue append rounds', ph[2], ps[4]), ('triangle batches (queue read, permutes, test, kill scatter)', ph[3], ps[5])]
    for name, a1, a2 in rows1:
        lines.append('| %s | %s | %s | %s | %.4g | %.1f %% |' % (name, '%.4g' % a1 if a1 is not None else '', '%.1f %%' % (100.0 * a1 / total1) if a1 is not None else '',
                                                                  '%.0f' % (a1 / max(iters, 1)) if a1 is not None else '', a2, 100.0 * a2 / total2))
    rest1, rest2 = total1 - ph[0] - ph[1] - ph[2] - ph[3], total2 - ps[0] - ps[1] - ps[2] - ps[3] - ps[4] - ps[5]
    lines += ['| loop control between the phases, prologue, epilogue, the clock reads themselves | %.4g | %.1f %% | %.0f | %.4g | %.1f %% |'
              % (rest1, 100.0 * rest1 / total1, rest1 / max(iters, 1), rest2, 100.0 * rest2 / total2),
              '| total | %.4g | | %.0f | %.4g | |' % (total1, total1 / max(iters, 1), total2), '',
              'loop iterations %d, of which with a node step in some lane %d (%.1f %%); lanes stepping per node-step iteration %.1f of 64; triangle batches %d at %.1f %% fill'
              % (iters, step_iters, 100.0 * step_iters / max(iters, 1), f.node_steps / max(step_iters, 1), f.leaf_batches[0],
                 100.0 * f.leaf_batches[1] / max(64 * f.leaf_batches[0], 1)), '',
              'timeline (counting launch, %.1f us from the first wavefront\'s begin to the last one\'s end at %.0f MHz): wavefronts begin at %.1f / %.1f / %.1f %% of the span '
              '(median / 90th / last), end at %.1f / %.1f / %.1f / 100 %% (10th / median / 90th / last); slots busy %.1f %% of span x wavefronts; '
              'the last 10 %% of the span holds %.1f %% of the wavefronts\' ends'
              % (span / 100.0, f.clock_mhz, 100 * q(begin, 0.5), 100 * q(begin, 0.9), 100 * float(begin.max()), 100 * q(end, 0.1), 100 * q(end, 0.5), 100 * q(end, 0.9),
                 100.0 * busy, 100.0 * float((end > 0.9).double().mean())), '', who, '']
    del st
    torch.cuda.empty_cache()
out = '\n'.join(lines)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(out + '\n')
