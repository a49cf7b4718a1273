"""GPU parity probe for the fused env-shade kernel against the CPU oracle (bob 128^2, several n)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _build, _lib
from oracle import oracle as orc, scene_cpu
lib = ctypes.CDLL(_build.LIB)
lib.nvdr_last_error.restype = ctypes.c_char_p
def chk(rc, what):
    assert rc == 0, (what, rc, lib.nvdr_last_error())
P = lambda t: ctypes.c_void_p(t.data_ptr())
S_ = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = torch.device('cuda:0')
ctx = ctypes.c_void_p()
chk(lib.nvdr_ctx_create(ctypes.byref(ctx), 0), 'ctx')

def gpu_shade(inp, n, bsdf, seed, dgrad=None, sgrad=None, vis_cache=None):
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items() if k != 'mesh'}
    a = _lib.NvdrEnvShadeArgs()
    a.mask = _lib.tensor_view(g['mask'], lead=False)
    for k in ('ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks'):
        setattr(a, k, _lib.tensor_view(g[k]))
    a.light = _lib.tensor_view(g['light'], lead=False); a.pdf = _lib.tensor_view(g['pdf'], lead=False)
    a.rows = _lib.tensor_view(g['rows'], lead=False); a.cols = _lib.tensor_view(g['cols'], lead=False)
    a.perms = _lib.tensor_view(g['perms'], lead=False)
    a.bsdf = bsdf; a.n_samples_x = n; a.rnd_seed = seed; a.shadow_scale = 1.0; a.pixel_index_offset = 0
    N, H, W = g['ro'].shape[:3]
    out = {}
    if vis_cache is not None: a.vis_cache = vis_cache.data_ptr()
    if dgrad is None:
        out['diff'] = torch.empty(N, H, W, 3, device=dev); out['spec'] = torch.empty(N, H, W, 3, device=dev)
        a.diff = out['diff'].data_ptr(); a.spec = out['spec'].data_ptr()
        chk(lib.nvdr_env_shade_fwd(ctx, ctypes.byref(a), S_()), 'fwd')
    else:
        dg, sg = dgrad.to(dev), sgrad.to(dev)
        a.diff_grad = _lib.tensor_view(dg); a.spec_grad = _lib.tensor_view(sg)
        for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad'):
            out[k] = torch.empty(N, H, W, 3, device=dev); setattr(a, k, out[k].data_ptr())
        out['light_grad'] = torch.empty(g['light'].shape[0], g['light'].shape[1], 3, device=dev); a.light_grad = out['light_grad'].data_ptr()
        chk(lib.nvdr_env_shade_bwd(ctx, ctypes.byref(a), S_()), 'bwd')
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}

def report(tag, a, b):
    d = (a - b).abs(); rel = d / (b.abs() + 1e-3)
    print('  %-16s max abs %.3e  max rel %.3e  n(rel>1e-4) %d / %d  mean|ref| %.3e' % (tag, d.max().item(), rel.max().item(), int((rel > 1e-4).sum()), d.numel(), b.abs().mean().item()))

for n, bsdf in ((2, 0), (3, 0), (8, 0), (2, 1)):
    inp = scene_cpu.make_inputs('bob', 128, 128, n)
    m = inp['mesh']
    v = m['v_pos'].to(dev).contiguous(); t = m['t_pos_idx'].to(dev).contiguous()
    chk(lib.nvdr_bvh_build(ctx, P(v), ctypes.c_int64(v.shape[0]), P(t), ctypes.c_int64(t.shape[0]), 1, S_()), 'build')
    kw = scene_cpu.shade_kwargs(inp)
    t0 = time.time()
    ref = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=7, n_threads=orc.max_threads())
    print('n=%d bsdf=%d covered=%d oracle fwd %.2fs' % (n, bsdf, ref['covered'], time.time() - t0))
    S = n * n; words = (S + 31) // 32
    vc = torch.zeros(128 * 128 * 2 * words, dtype=torch.int32, device=dev)
    got = gpu_shade(inp, n, bsdf, 7, vis_cache=vc)
    report('diff', got['diff'], ref['diff']); report('spec', got['spec'], ref['spec'])
    g = torch.Generator().manual_seed(3)
    dg = torch.rand(1, 128, 128, 3, generator=g); sg = torch.rand(1, 128, 128, 3, generator=g)
    refb = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf=bsdf, n_samples_x=n, rnd_seed=7, diff_grad=dg, spec_grad=sg, n_threads=orc.max_threads())
    gotb = gpu_shade(inp, n, bsdf, 7, dg, sg)
    gotc = gpu_shade(inp, n, bsdf, 7, dg, sg, vis_cache=vc)
    for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad', 'light_grad'):
        report(k, gotb[k], refb[k])
        print('     cached-vis bwd == traced bwd:', bool((gotb[k] - gotc[k]).abs().max().item() < (1e-5 if k != 'light_grad' else 1e-3) * (1 + refb[k].abs().max().item())))
print('OK')
