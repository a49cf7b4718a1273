"""A/B of library variants INSIDE ONE PROCESS: every variant of libnvdr_hip.so under csrc/build/variants/ (and the current build)
is loaded side by side (separate dlopen handles, separate globals), gets its own DirectLightingStep, and the variants are timed
in interleaved rounds -- fresh processes differ by +-8 % on identical code, which buries most kernel experiments.
    PROBE_VIEWS=8 python tools/ab_inproc.py [rounds]"""
import os as _os; _os.environ.setdefault('NVDR_TUNING', '1')
import ctypes, glob, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _lib, _build
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru
from tools.gpu_tenancy import snapshot

res = int(os.environ.get('PROBE_RES', '512'))
subdiv = int(os.environ.get('PROBE_SUBDIV', '0'))
nviews = int(os.environ.get('PROBE_VIEWS', '8'))
n_x = int(os.environ.get('PROBE_N', '8'))                # n_samples_x
mesh_name = os.environ.get('PROBE_MESH', 'bob')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
base = _build.LIB
paths = [('current', base)] + [(p.split('.so.')[-1], p) for p in sorted(glob.glob(os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.*')))]
# AB_ENV="tag:VAR=value;VAR2=value|tag2:VAR=value": the current library with environment switches set while the context is created
env_variants = {}
for spec in filter(None, os.environ.get('AB_ENV', '').split('|')):
    tag, kv = spec.split(':', 1)
    env_variants[tag] = dict(x.split('=', 1) for x in kv.split(';'))
    paths.append((tag, base))
if os.environ.get('AB_LG', '0') != '0':
    paths.append(('lg_all', base))         # light-gradient gather: every workgroup walks all bands (NVDR_LG_MODE=0)
    paths.append(('lg_perband', base))     # one set of workgroups per band (NVDR_LG_MODE=1)
only = os.environ.get('AB_ONLY')
if only:
    paths = [pp for pp in paths if pp[0] in only.split(',') or pp[0] == 'current']

steps = {}
for tag, path in paths:
    _lib._lib = None
    _build.LIB = path                      # _lib.load() binds the signatures of whatever this points to
    lib = _lib.load()
    for k_, v_ in env_variants.get(tag, {}).items():
        os.environ[k_] = v_
    if tag.startswith('lg_'):
        os.environ['NVDR_LG_MODE'] = '1' if tag == 'lg_perband' else '0'
    st = DirectLightingStep(mesh_name, res, n_x, view=list(range(nviews)), n_views=8, device='cuda:0', subdiv=subdiv, retrace_backward=True)
    os.environ.pop('NVDR_LG_MODE', None)
    for k_ in env_variants.get(tag, {}):
        os.environ.pop(k_, None)
    assert st.ctx.cpp_wrapper.lib is lib
    with torch.no_grad():
        m = st.mask[..., None]
        _, ro, _, nrm, _, kd, ks = st.shade_inputs()
    steps[tag] = (st, kd, ks, nrm, ro)
_build.LIB = base


def run(tag, iters=int(os.environ.get('AB_ITERS', '4'))):
    """Average stage times (gen, trace, shade forward; trace, shade backward) over iters - 1 forward + backward passes."""
    st, kd, ks, nrm, ro = steps[tag]
    L = st.light
    st.ctx.set_profiling(True)
    for it in range(iters):
        if it == 1:
            st.ctx.set_profiling(True)
        g = [t.clone().requires_grad_(True) for t in (st.gb_pos, nrm, kd, ks, L.base.detach())]
        d, s = ou.optix_env_shade(st.ctx, st.mask, ro, g[0], g[1], st.view_pos, g[2], g[3], g[4], L._pdf, L.rows[:, 0], L.cols,
                                  n_samples_x=n_x, rnd_seed=it, shadow_scale=1.0)
        torch.autograd.backward([d, s], [torch.ones_like(d), torch.ones_like(s)])
    torch.cuda.synchronize()
    nf, f = st.ctx.stage_times(backward=False)
    nb, b = st.ctx.stage_times(backward=True)
    st.ctx.set_profiling(False)
    return [f[0], f[1], f[2], b[1], b[2]]


ou.ops._optix_env_shade_func.cache_visibility = False          # backward re-traces, as the benchmark does
for tag in steps:
    run(tag, 2)                            # warm every variant
times = {tag: [] for tag in steps}
for r in range(rounds):
    for tag in steps:
        times[tag].append(run(tag))
names = ['gen', 'trace', 'shade', 'bwd trace', 'bwd shade+gather']
med = {tag: [statistics.median(v[k] for v in times[tag]) for k in range(5)] for tag in steps}
print('env-shade stage times, %d views, %d rounds interleaved in one process: median ms (vs current)' % (nviews, rounds))
print('  %-10s %s' % ('', '  '.join('%-18s' % n for n in names)))
for tag in steps:
    print('  %-10s %s' % (tag, '  '.join('%7.3f (%+5.1f %%)  ' % (med[tag][k], 100.0 * (med[tag][k] / med['current'][k] - 1.0)) for k in range(5))))
print('traversal ms of every measured pass (forward | backward):')
for tag in steps:
    print('  %-10s %s | %s' % (tag, ' '.join('%.1f' % v[1] for v in times[tag]), ' '.join('%.1f' % v[3] for v in times[tag])))
print('tenancy at the end: ' + snapshot())
