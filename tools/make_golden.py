#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself in this container.

  renderutils_*.npz : outputs + gradients of the reference's pure-PyTorch operators
                      (render/renderutils/{bsdf,loss}.py via ops.py use_python=True), imported from
                      /root/reference on the CPU, with the input ranges of the reference's own tests
                      (render/renderutils/tests/test_{bsdf,loss,mesh}.py).
  env_shade_*.npz   : forward + backward of the reference's raygen program (envsampling/kernel.cu) compiled
                      for the CPU by oracle/Makefile `ref` (two builds: libm transcendentals, and the
                      detmath build whose discrete decisions our kernels must reproduce bit for bit).
  denoiser_*.npz    : forward + backward of the reference's denoising.cu kernels, same route.
  light_reference.npz : _pdf / cols / rows of the reference's own EnvironmentLight.update_pdf (render/light.py:46-59),
                      imported from /root/reference with its two CUDA-only imports (nvdiffrast, imageio) stubbed in
                      sys.modules and torch.arange stripped of device="cuda" (render/util.py:62-66 builds the pixel
                      grid on 'cuda'); probes E0 / E1 at 256x256 and 48x48, plus a probe with an all-zero row.

  dmtet_reference.npz : the reference's own marching_tets / map_uv (geometry/dmtet.py:50-141) on an own 6^3 Kuhn tet grid with two
                      seeded SDFs: the pin of the numpy restatement that extracts the DMTet benchmark meshes.

Inputs of the env-shade / denoiser cases are NOT stored (they are regenerated from seeds by
oracle/scene_cpu.py); a checksum of them is stored and re-checked by the tests.

    python tools/make_golden.py      # needs /root/reference; /root/reference is never written to
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'


def checksum(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def gen_renderutils():
    sys.path.insert(0, os.path.join(REF, 'render'))
    import renderutils as ru  # the reference module (lazy plugin: pure python paths run on the CPU)
    g = torch.Generator().manual_seed(1234)
    R = lambda *s, lo=0.0, hi=1.0: (torch.rand(*s, generator=g) * (hi - lo) + lo)
    cases = {}

    def run(name, fn, inputs, out_ch):
        ins = [t.clone().requires_grad_(True) for t in inputs]
        out = fn(*ins)
        target = R(*out.shape)
        loss = torch.nn.functional.mse_loss(out, target)
        loss.backward()
        d = {'n_in': np.int32(len(ins)), 'out': out.detach().numpy(), 'target': target.numpy()}
        for i, t in enumerate(ins):
            d['in%d' % i] = t.detach().numpy()
            d['grad%d' % i] = t.grad.numpy()
        cases[name] = d

    RES = 4
    s3, s1 = (1, RES, RES, 3), (1, RES, RES, 1)
    # prepare_shading_normal (test_bsdf.py:24-56), both flag combinations
    for two, ogl in ((True, True), (False, False)):
        run('prepare_shading_normal_%d%d' % (two, ogl),
            lambda p, v, pn, sn, st, gn: ru.prepare_shading_normal(p, v, pn, sn, st, gn, two_sided_shading=two, opengl=ogl, use_python=True),
            [R(*s3), R(*s3), R(*s3), R(*s3), R(*s3), R(*s3)], 3)
    # broadcast view_pos [1,1,1,3] as render.py uses it
    run('prepare_shading_normal_bcast',
        lambda p, v, pn, sn, st, gn: ru.prepare_shading_normal(p, v, pn, sn, st, gn, use_python=True),
        [R(*s3, lo=-1, hi=1), R(1, 1, 1, 3, lo=2, hi=3), R(1, 1, 1, 3), R(*s3, lo=-1, hi=1), R(*s3, lo=-1, hi=1), R(*s3, lo=-1, hi=1)], 3)
    run('fresnel_shlick', lambda a, b, c: ru._fresnel_shlick(a, b, c, use_python=True), [R(*s3), R(*s3), R(*s1, hi=2.0)], 3)
    run('ndf_ggx', lambda a, c: ru._ndf_ggx(a, c, use_python=True), [R(*s1), R(*s1, lo=-1.0, hi=2.0)], 1)
    run('lambda_ggx', lambda a, c: ru._lambda_ggx(a, c, use_python=True), [R(*s1), R(*s1, lo=-1.0, hi=2.0)], 1)
    run('masking_smith', lambda a, i, o: ru._masking_smith(a, i, o, use_python=True), [R(*s1), R(*s1), R(*s1)], 1)
    nrm = lambda: torch.nn.functional.normalize(R(*s3), dim=-1)
    run('lambert', lambda n, w: ru.lambert(n, w, use_python=True), [nrm(), nrm()], 1)
    run('frostbite', lambda n, wi, wo, r: ru.frostbite_diffuse(n, wi, wo, r, use_python=True), [nrm(), nrm(), nrm(), R(*s1)], 1)
    run('pbr_specular', lambda c, n, wo, wi, a: ru.pbr_specular(c, n, wo, wi, a, use_python=True), [R(*s3), nrm(), nrm(), nrm(), R(*s1)], 3)
    for b in ('lambert', 'frostbite'):
        run('pbr_bsdf_' + b, lambda kd, arm, pos, n, v, l: ru.pbr_bsdf(kd, arm, pos, n, v, l, bsdf=b, use_python=True),
            [R(*s3), R(*s3), R(*s3), nrm(), R(*s3), R(*s3)], 3)
    # image loss (test_loss.py:55-60) + n2n
    L = 8
    for loss, tm in (('l1', 'none'), ('l1', 'log_srgb'), ('mse', 'log_srgb'), ('smape', 'none'), ('relmse', 'none'), ('mse', 'none'), ('n2n', 'none')):
        ins = [R(1, L, L, 3).requires_grad_(True), R(1, L, L, 3).requires_grad_(True)]
        out = ru.image_loss(ins[0], ins[1], loss=loss, tonemapper=tm, use_python=True)
        out.backward()
        cases['image_loss_%s_%s' % (loss, tm)] = {'n_in': np.int32(2), 'out': out.detach().numpy(), 'in0': ins[0].detach().numpy(),
                                                  'in1': ins[1].detach().numpy(), 'grad0': ins[0].grad.numpy(), 'grad1': ins[1].grad.numpy()}
    # xfm (test_mesh.py:37-86): 8 matrices, 1024 points, shared and per-batch points
    for name, fn, pb in (('xfm_points', ru.xfm_points, 1), ('xfm_vectors', ru.xfm_vectors, 1), ('xfm_points_b', ru.xfm_points, 8)):
        run(name, lambda p, m: fn(p, m, use_python=True), [R(pb, 1024, 3), R(8, 4, 4)], 4)
    flat = {}
    for k, d in cases.items():
        for kk, v in d.items():
            flat[k + '/' + kk] = v
    np.savez_compressed(os.path.join(OUT, 'renderutils_reference.npz'), **flat)
    print('renderutils: %d cases' % len(cases))


ENV_CASES = [
    # name, mesh, H, W, n, bsdf, seed, probe_res, env
    ('bob_pbr_n2', 'bob', 40, 48, 2, 'pbr', 11, 64, 'E1'),
    ('bob_pbr_n3', 'bob', 32, 32, 3, 'pbr', 5, 64, 'E1'),
    ('bob_diffuse_n2', 'bob', 32, 32, 2, 'diffuse', 3, 64, 'E0'),
    ('spot_metal_n2', 'spot', 32, 32, 2, 'pbr', 9, 64, 'E1'),
]


def env_case_inputs(mesh, H, W, n, probe_res, env):
    from oracle import scene_cpu
    return scene_cpu.make_inputs(mesh, H, W, n, probe_res=probe_res, env=env, n_threads=8)


def gen_env_shade():
    from oracle import oracle as orc, scene_cpu
    assert orc.build_ref() is not None and orc.have_ref()
    flat = {}
    for name, mesh, H, W, n, bsdf, seed, pr, env in ENV_CASES:
        inp = env_case_inputs(mesh, H, W, n, pr, env)
        kw = scene_cpu.shade_kwargs(inp)
        m = inp['mesh']
        g = torch.Generator().manual_seed(seed)
        dg, sg = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
        flat[name + '/inputs_sha256'] = np.array(checksum(*[kw[k] for k in sorted(kw)], dg, sg))
        for impl in ('ref', 'ref_detmath'):
            f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, n_threads=8, want_vis=True, impl=impl)
            b = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=seed, diff_grad=dg, spec_grad=sg, n_threads=1, impl=impl)
            for k in ('diff', 'spec'):
                flat['%s/%s/%s' % (name, impl, k)] = f[k].numpy()
            flat['%s/%s/vis' % (name, impl)] = np.packbits(f['vis'].numpy())
            for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad', 'light_grad'):
                flat['%s/%s/%s' % (name, impl, k)] = b[k].numpy()
        print('env_shade', name, 'covered', f['covered'])
    np.savez_compressed(os.path.join(OUT, 'env_shade_reference.npz'), **flat)


DN_CASES = [('s2', 1, 40, 36, 2.0, 21), ('s07', 2, 24, 20, 0.7, 22), ('s0', 1, 16, 16, 0.0001, 23)]


def denoiser_inputs(N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, H, W, 8, generator=g)
    col = x[..., 0:3]
    nrm = torch.nn.functional.normalize(x[..., 3:6] * 2 - 1, dim=-1)
    zdz = x[..., 6:8]
    og = torch.rand(N, H, W, 4, generator=g)
    return x, col, nrm, zdz, og


def gen_denoiser():
    from oracle import oracle as orc
    flat = {}
    for name, N, H, W, sigma, seed in DN_CASES:
        x, col, nrm, zdz, og = denoiser_inputs(N, H, W, seed)
        flat[name + '/inputs_sha256'] = np.array(checksum(x, nrm, og))
        flat[name + '/out'] = orc.bilateral_fwd(col, nrm, zdz, sigma, n_threads=8, impl='ref').numpy()
        flat[name + '/col_grad'] = orc.bilateral_bwd(col, nrm, zdz, sigma, og, n_threads=8, impl='ref').numpy()
    np.savez_compressed(os.path.join(OUT, 'denoiser_reference.npz'), **flat)
    print('denoiser: %d cases' % len(DN_CASES))


LIGHT_CASES = [('E0_256', 'E0', 256), ('E1_256', 'E1', 256), ('E0_48', 'E0', 48), ('E1_48', 'E1', 48), ('E1_64_zero_rows', 'E1', 64)]


def light_case_base(kind, res, name):
    from nvdiffrecmc_amd import scene as sc
    base = sc.env_map(kind, res).clone()
    if name.endswith('zero_rows'):
        base[5] = 0.0           # a row without light: its column CDF keeps the un-normalised zeros (light.py:58 `where`)
        base[-1] = 0.0
    return base


def gen_light():
    """The reference's EnvironmentLight.update_pdf itself: 12 lines of torch behind CUDA-only imports."""
    import types
    import importlib
    for mod in ('nvdiffrast', 'nvdiffrast.torch', 'imageio'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules['nvdiffrast'].torch = sys.modules['nvdiffrast.torch']
    if REF not in sys.path:
        sys.path.insert(0, REF)
    real_arange = torch.arange

    def cpu_arange(*a, **k):
        k.pop('device', None)
        return real_arange(*a, **k)
    torch.arange = cpu_arange
    try:
        light = importlib.import_module('render.light')     # /root/reference/render/light.py, unmodified
        flat = {}
        for name, kind, res in LIGHT_CASES:
            base = light_case_base(kind, res, name)
            lgt = light.EnvironmentLight(base)               # __init__ calls update_pdf (light.py:31-32)
            flat[name + '/base_sha256'] = np.array(checksum(base))
            flat[name + '/pdf'] = lgt._pdf.numpy()
            flat[name + '/cols'] = lgt.cols.numpy()
            flat[name + '/rows'] = lgt.rows.numpy()          # [H,W] with identical columns; render.py:114 passes rows[:,0]
            print('light', name, tuple(lgt._pdf.shape), float(lgt._pdf.sum()))
    finally:
        torch.arange = real_arange
    np.savez_compressed(os.path.join(OUT, 'light_reference.npz'), **flat)


def mesh_case(name='spot', n_tris=700):
    """A small sub-mesh (the first n_tris triangles of an asset, vertices compacted in both index spaces) + fixed upstream
    gradients: the input of the mesh-frame pins."""
    from nvdiffrecmc_amd import scene as sc
    m = sc.load_mesh(name)
    tp, tt = m['t_pos_idx'][:n_tris].long(), m['t_tex_idx'][:n_tris].long()
    up, ip = torch.unique(tp, return_inverse=True)
    ut, it = torch.unique(tt, return_inverse=True)
    g = torch.Generator().manual_seed(7)
    v_pos = m['v_pos'][up].contiguous()
    return {'v_pos': v_pos, 't_pos_idx': ip.int().contiguous(), 'v_tex': m['v_tex'][ut].contiguous(), 't_tex_idx': it.int().contiguous(),
            'g_nrm': torch.randn(v_pos.shape, generator=g), 'g_tng': torch.randn(v_pos.shape, generator=g)}


def gen_mesh():
    """The reference's own auto_normals / compute_tangents (render/mesh.py:150-219, imported unmodified with the CUDA-only imports
    stubbed and device='cuda' stripped from torch.tensor) and torch autograd through them: values and the gradient w.r.t. v_pos."""
    import types
    import importlib
    for mod in ('nvdiffrast', 'nvdiffrast.torch', 'imageio', 'tinycudann'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules['nvdiffrast'].torch = sys.modules['nvdiffrast.torch']
    if REF not in sys.path:
        sys.path.insert(0, REF)
    real_tensor = torch.tensor

    def cpu_tensor(*a, **k):
        k.pop('device', None)
        return real_tensor(*a, **k)
    torch.tensor = cpu_tensor
    try:
        rmesh = importlib.import_module('render.mesh')      # /root/reference/render/mesh.py, unmodified
        c = mesh_case()
        v_pos = c['v_pos'].clone().requires_grad_(True)
        im = rmesh.Mesh(v_pos=v_pos, t_pos_idx=c['t_pos_idx'].long(), v_tex=c['v_tex'], t_tex_idx=c['t_tex_idx'].long())
        im = rmesh.auto_normals(im)
        im = rmesh.compute_tangents(im)
        ((im.v_nrm * c['g_nrm']).sum() + (im.v_tng * c['g_tng']).sum()).backward()
        flat = {'spot700/v_pos_sha256': np.array(checksum(c['v_pos'], c['t_pos_idx'], c['v_tex'], c['t_tex_idx'], c['g_nrm'], c['g_tng'])),
                'spot700/v_nrm': im.v_nrm.detach().numpy(), 'spot700/v_tng': im.v_tng.detach().numpy(),
                'spot700/v_pos_grad': v_pos.grad.numpy()}
        assert torch.equal(im.t_nrm_idx, im.t_pos_idx) and torch.equal(im.t_tng_idx, im.t_pos_idx)     # mesh.py:178,219
        print('mesh', tuple(im.v_nrm.shape), float(v_pos.grad.abs().max()))
    finally:
        torch.tensor = real_tensor
    np.savez_compressed(os.path.join(OUT, 'mesh_reference.npz'), **flat)


def kuhn_tet_grid(k):
    """An own small tet grid for the marching-tets pin (NOT reference data): the unit cube cut into k^3 cells of six Kuhn tetrahedra each."""
    ax = np.arange(k + 1)
    vid = lambda x, y, z: (x * (k + 1) + y) * (k + 1) + z
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    verts = (np.stack((X, Y, Z), -1).reshape(-1, 3) / float(k) - 0.5).astype(np.float32)
    import itertools
    tets = []
    for x in range(k):
        for y in range(k):
            for z in range(k):
                for perm in itertools.permutations(range(3)):
                    p = [x, y, z]
                    tet = [vid(*p)]
                    for a in perm:
                        p[a] += 1
                        tet.append(vid(*p))
                    tets.append(tet)
    return verts, np.asarray(tets, dtype=np.int64)


def gen_dmtet():
    """The reference's own marching_tets / map_uv (geometry/dmtet.py:50-141, the module imported unmodified with its renderer imports
    stubbed and device='cuda' stripped from the tensor factories) on an own 6^3 Kuhn grid with two seeded SDFs: the pin of
    tools/make_dmtet_mesh.py's numpy restatement and of nvdiffrecmc_amd/scene.py's atlas."""
    import types
    import importlib
    for mod in ('nvdiffrast', 'nvdiffrast.torch', 'imageio', 'tinycudann', 'render.render', 'render.regularizer', 'render.optixutils'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules['nvdiffrast'].torch = sys.modules['nvdiffrast.torch']
    if REF not in sys.path:
        sys.path.insert(0, REF)
    real = {n: getattr(torch, n) for n in ('tensor', 'ones', 'arange', 'linspace')}

    def strip(fn):
        def f(*a, **k):
            k.pop('device', None)
            return fn(*a, **k)
        return f
    for n, fn in real.items():
        setattr(torch, n, strip(fn))
    try:
        import render as _render_pkg
        for sub in ('render', 'regularizer', 'optixutils'):
            setattr(_render_pkg, sub, sys.modules['render.' + sub])
        dm = importlib.import_module('geometry.dmtet')       # /root/reference/geometry/dmtet.py, unmodified
        verts, tets = kuhn_tet_grid(6)
        flat = {'grid/vertices': verts, 'grid/indices': tets}
        rng = np.random.default_rng(99)
        sdfs = {'random': rng.random(verts.shape[0]).astype(np.float32) - np.float32(0.1),
                'sphere': (0.37 - np.linalg.norm(verts + 0.03, axis=-1)).astype(np.float32)}
        for name, sdf in sdfs.items():
            v, f, uvs, uv_idx = dm.marching_tets(torch.from_numpy(verts) * 2.4, torch.from_numpy(sdf), torch.from_numpy(tets))
            flat[name + '/sdf'] = sdf
            flat[name + '/verts'] = v.numpy()
            flat[name + '/faces'] = f.numpy().astype(np.int32)
            flat[name + '/uv_idx'] = uv_idx.numpy().astype(np.int32)
            flat[name + '/uvs_sha256'] = np.array(checksum(uvs))
            flat[name + '/uvs_head'] = uvs[:64].numpy()
            print('dmtet', name, tuple(v.shape), tuple(f.shape), tuple(uvs.shape))
    finally:
        for n, fn in real.items():
            setattr(torch, n, fn)
    np.savez_compressed(os.path.join(OUT, 'dmtet_reference.npz'), **flat)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'dmtet':
        gen_dmtet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'light':
        gen_light()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'mesh':
        gen_mesh()
        sys.exit(0)
    gen_renderutils()
    gen_light()
    gen_mesh()
    gen_dmtet()
    gen_env_shade()
    gen_denoiser()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
