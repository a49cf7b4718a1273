"""Bilateral filter forward / backward times of every library variant under csrc/build/variants/ (and the current build), in one
process, interleaved rounds: PROBE_VIEWS=8 python tools/dn_probe.py [rounds]"""
import glob, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _lib, _build
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd.denoiser import _safe_normalize
from nvdiffrecmc_amd import optixutils as ou, renderutils as ru

res = int(os.environ.get('PROBE_RES', '512'))
nviews = int(os.environ.get('PROBE_VIEWS', '8'))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
base = _build.LIB
paths = [('current', base)] + [(p.split('.so.')[-1], p) for p in sorted(glob.glob(os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.*')))]
only = os.environ.get('AB_ONLY')
if only:
    paths = [pp for pp in paths if pp[0] in only.split(',') or pp[0] == 'current']
_build.LIB = base
_lib._lib = None
st = DirectLightingStep('bob', res, 8, view=list(range(nviews)), n_views=8, device='cuda:0', retrace_backward=True)
with torch.no_grad():
    nrm = ru.prepare_shading_normal(st.gb_pos, st.view_pos, None, st.gb_smooth_nrm, st.gb_tangent, st.gb_geom_nrm)
    nn = _safe_normalize(nrm).contiguous()
    col = (torch.rand_like(nn) * st.mask[..., None]).contiguous()
    og = torch.rand(st.nv, res, res, 4, device=col.device)
libs = {}
for tag, path in paths:
    _lib._lib = None
    _build.LIB = path
    libs[tag] = _lib.load()
_build.LIB = base
sigma = st.denoiser.sigma


def run(tag, iters=10):
    _lib._lib = libs[tag]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    c = col.clone().requires_grad_(True)
    out = None
    tf = tb = 0.0
    for it in range(iters):
        c.grad = None
        ev[0].record()
        out = ou.ops._bilateral_denoiser_func.apply(c, nn, st.gb_depth, sigma)
        ev[1].record()
        out.backward(og)
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / (iters - 1), tb / (iters - 1), out.detach(), c.grad.detach()


ref = run('current', 2)
acc = {t: [] for t, _ in paths}
for r in range(rounds):
    for tag, _ in paths:
        f, b, o, g = run(tag)
        acc[tag].append((f, b))
        assert torch.equal(o, ref[2]) and torch.equal(g, ref[3]), 'variant %s changes the filter output' % tag


def run_pair(tag='current', iters=10):
    _lib._lib = libs[tag]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    c1, c2 = col.clone().requires_grad_(True), (col * 0.5).clone().requires_grad_(True)
    tf = tb = 0.0
    for it in range(iters):
        c1.grad = c2.grad = None
        ev[0].record()
        o1, o2 = ou.ops._bilateral_denoiser_pair_func.apply(c1, c2, nn, st.gb_depth, sigma)
        ev[1].record()
        torch.autograd.backward([o1, o2], [og, og])
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / (iters - 1), tb / (iters - 1)


if hasattr(ou.ops, '_bilateral_denoiser_pair_func'):
    acc_p = {t: [] for t, _ in paths}
    for r in range(rounds):
        for t, _ in paths:
            acc_p[t].append(run_pair(t))
    for t, _ in paths:
        print('pair kernel (two images in one pass) %-10s fwd %.3f  bwd %.3f   (two single calls = twice the figures below)' % (t, statistics.median(x[0] for x in acc_p[t]), statistics.median(x[1] for x in acc_p[t])))
print('bilateral filter, %d views %dx%d, sigma %g: median ms over %d interleaved rounds (outputs bit-identical)' % (nviews, res, res, sigma, rounds))
for tag, _ in paths:
    print('  %-10s fwd %.3f  bwd %.3f' % (tag, statistics.median(x[0] for x in acc[tag]), statistics.median(x[1] for x in acc[tag])))
