"""Bit-compare the env-shade outputs (images and per-pixel gradients) of a library variant with the current build, in two processes.
    python tools/variant_equal.py <variant tag> [n_samples_x ...]      (variants: tools/build_variants.sh)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
from nvdiffrecmc_amd import _build
if %(lib)r:
    _build.LIB = %(lib)r
from nvdiffrecmc_amd.trainer import DirectLightingStep
from nvdiffrecmc_amd import optixutils as ou
out = {}
for n in %(ns)r:
    st = DirectLightingStep('bob', 256, n, view=[0, 1], n_views=8, device='cuda:0', retrace_backward=True)
    with torch.no_grad():
        _, ro, _, nrm, _, kd, ks = st.shade_inputs()
    L = st.light
    g = [t.clone().requires_grad_(True) for t in (st.gb_pos, nrm, kd, ks)]
    d, s = ou.optix_env_shade(st.ctx, st.mask, ro, g[0], g[1], st.view_pos, g[2], g[3], L.base.detach(), L._pdf, L.rows[:, 0], L.cols,
                              n_samples_x=n, rnd_seed=3, shadow_scale=1.0)
    torch.autograd.backward([d, s], [torch.ones_like(d), torch.ones_like(s)])
    out[n] = [d.detach().cpu(), s.detach().cpu()] + [t.grad.cpu() for t in g]
torch.save(out, %(path)r)
'''


def run(lib, ns, path):
    code = CHILD % {'root': ROOT, 'lib': lib, 'ns': ns, 'path': path}
    env = dict(os.environ, NVDR_TUNING='1')
    subprocess.run([sys.executable, '-c', code], check=True, env=env, stderr=subprocess.DEVNULL)


def main():
    import torch
    tag = sys.argv[1]
    ns = [int(a) for a in sys.argv[2:]] or [8, 4, 16]
    from nvdiffrecmc_amd import _build
    var = os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.' + tag)
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, 'a.pt'), os.path.join(tmp, 'b.pt')
        run('', ns, a)
        run(var, ns, b)
        A, B = torch.load(a), torch.load(b)
    names = ['diffuse', 'specular', 'd gb_pos', 'd gb_normal', 'd kd', 'd ks']
    for n in ns:
        print('n_samples_x = %2d:' % n, ', '.join('%s %s' % (nm, 'equal' if torch.equal(x, y) else 'DIFFERENT (max |delta| %.3e)' % float((x - y).abs().max()))
                                                    for nm, x, y in zip(names, A[n], B[n])))


if __name__ == '__main__':
    main()
