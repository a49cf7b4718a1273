"""What would it cost in accuracy to evaluate the VALUE-ONLY arithmetic of the raygen program the fast way on the GPU (a * rcp(b)
with a 1-ulp reciprocal instead of IEEE division, float instead of the fp64 islands the reference inherits from CUDART_PI and
unsuffixed literals), while everything that feeds a discrete decision (directions, texels, lobe choice, dead-sample gate, CDF
inversion) stays exact?  Answered on the CPU: oracle/nvdr_oracle.c is built twice (the checker build, and -DORACLE_FAST_VALUE_MATH=1,
which perturbs every such reciprocal by up to one ulp) and the two are compared with the metrics of the -m gpu parity tests
(tests/test_gpu_env_shade.py: forward |d| <= 2e-6 (|ref| + 1e-3), per-pixel gradients 2e-4 (|ref| + 1e-3 max|ref|), light 1e-4).
    python tools/fast_value_math.py            -> table on stdout (committed as profiles/r03_fast_value_math.md)"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, scene_cpu  # noqa: E402

def build_fast(level):
    src = os.path.join(ROOT, 'oracle', 'nvdr_oracle.c')
    out = os.path.join(ROOT, 'oracle', '_build', 'libnvdr_oracle_fast%d.so' % level)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-mfma', '-mavx2', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-fPIC', '-shared', '-Wno-unused-function',
                               '-I' + os.path.join(ROOT, 'include'), '-std=gnu11', '-DORACLE_FAST_VALUE_MATH=%d' % level, src, '-o', out, '-lm'])
    return out


def run(lib, m, kw, bsdf, n, seed, dg, sg, nt):
    exact = orc.LIB
    orc.LIB = lib
    try:
        f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, n_threads=nt, want_vis=True)
        b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=1, vis_in=f['vis'])
    finally:
        orc.LIB = exact
    return f, b


def main():
    orc.build()
    levels = ((1, 'LEVEL 1: pdfs and the MIS weight only'), (2, 'LEVEL 2: + BSDF evaluation and its adjoints'),
              (3, 'LEVEL 3 (round 5): FAITHFUL float divisions (quotient off by <= 1 ulp) in the BSDF evaluation, its adjoints and the MIS weight; fp64 islands kept'),
              (4, 'LEVEL 4 (round 5): level 3 + the pdfs of stage 1'))
    if len(sys.argv) > 1:
        levels = tuple(l for l in levels if str(l[0]) in sys.argv[1:])
    for level, what in levels:
        print('\n### %s\n' % what)
        table(build_fast(level))


def table(FAST):
    nt = orc.max_threads()
    cases = [('bob', 64, 8, 'pbr', 0, 3), ('bob', 64, 8, 'pbr', 5, 4), ('spot', 64, 8, 'pbr', 2, 5), ('spot', 48, 16, 'pbr', 6, 6),
             ('bob', 64, 4, 'diffuse', 1, 7), ('bob', 96, 8, 'pbr', 3, 8)]
    print('| scene | covered px | vis bits equal | fwd diff | fwd spec | d gb_pos | d gb_normal | d gb_kd | d gb_ks | d light |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    worst = {}
    for mesh, res, n, bsdf, view, seed in cases:
        inp = scene_cpu.make_inputs(mesh, res, res, n, view=view, probe_res=128, n_threads=nt)
        kw = scene_cpu.shade_kwargs(inp)
        m = inp['mesh']
        g = torch.Generator().manual_seed(seed)
        dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
        fe, be = run(orc.LIB, m, kw, bsdf, n, seed, dg, sg, nt)
        ff, bf = run(FAST, m, kw, bsdf, n, seed, dg, sg, nt)
        row = []
        for k in ('diff', 'spec'):
            e = ((ff[k] - fe[k]).abs() / (fe[k].abs() + 1e-3)).max().item() / 2e-6
            row.append(e)
            worst[k] = max(worst.get(k, 0), e)
        for k, tol in (('gb_pos_grad', 2e-4), ('gb_normal_grad', 2e-4), ('gb_kd_grad', 2e-4), ('gb_ks_grad', 2e-4), ('light_grad', 1e-4)):
            ref = be[k]
            floor = 1e-3 * max(1.0, ref.abs().max().item())
            e = ((bf[k] - ref).abs() / (ref.abs() + floor)).max().item() / tol
            row.append(e)
            worst[k] = max(worst.get(k, 0), e)
        print('| %s %dx%d n=%d %s view %d | %d | %s | %s |' % (mesh, res, res, n, bsdf, view, fe['covered'], bool(torch.equal(fe['vis'], ff['vis'])),
                                                           ' | '.join('%.3f' % v for v in row)))
    print('\nEntries are the largest error of the fast build against the checker build IN UNITS OF THE TEST TOLERANCE of that quantity '
          '(1.0 = the tolerance the GPU parity tests allow in total).  Worst over all scenes: ' + ', '.join('%s %.3f' % kv for kv in worst.items()))


if __name__ == '__main__':
    main()
