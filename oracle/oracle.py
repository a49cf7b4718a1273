"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/nvdr_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (nvdiffrecmc_amd/) never does.  Everything here runs on HOST memory (CPU torch
tensors / numpy arrays).

  build()            compile oracle/nvdr_oracle.c  -> oracle/_build/libnvdr_oracle.so      (gcc)
  build_ref()        compile the REFERENCE's own device sources for the CPU -> oracle/_ref/  (g++,
                     only where /root/reference exists; see oracle/Makefile, oracle/ref_shim/)
  env_shade(...)     fwd / bwd of the raygen program with brute-force visibility
  bilateral_*(...)   denoiser fwd / bwd
  visibility / closest / light_update_pdf
  have_ref() / ref_env_shade / ref_bilateral_*   the same entry points served by oracle/_ref
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libnvdr_oracle.so')
REF_LIB = os.path.join(HERE, '_ref', 'libnvdr_ref.so')
REF_DM_LIB = os.path.join(HERE, '_ref', 'libnvdr_ref_detmath.so')
REFERENCE_ROOT = '/root/reference'

c_void_p, c_int, c_long, c_float, c_uint32, c_int64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float,
                                                       ctypes.c_uint32, ctypes.c_int64)


class Tensor(ctypes.Structure):  # mirrors nvdr_tensor (include/nvdr_hip.h)
    _fields_ = [('data', c_void_p), ('size', c_int64 * 4), ('stride', c_int64 * 4)]


class EnvShadeArgs(ctypes.Structure):  # mirrors nvdr_env_shade_args
    _fields_ = [(n, Tensor) for n in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks',
                                      'light', 'pdf', 'rows', 'cols', 'perms')] + [
        ('bsdf', c_uint32), ('n_samples_x', c_uint32), ('rnd_seed', c_uint32), ('shadow_scale', c_float),
        ('pixel_index_offset', c_uint32),
        ('diff', c_void_p), ('spec', c_void_p),
        ('diff_grad', Tensor), ('spec_grad', Tensor),
        ('gb_pos_grad', c_void_p), ('gb_normal_grad', c_void_p), ('gb_kd_grad', c_void_p), ('gb_ks_grad', c_void_p),
        ('light_grad', c_void_p), ('vis_cache', c_void_p), ('counters', c_void_p), ('reuse_stream_id', ctypes.c_uint64), ('rnd_seed_offset', c_void_p),
        ('rnd_seed_snapshot', c_void_p), ('rnd_seed_advance', ctypes.c_uint32), ('phase', ctypes.c_uint32)]


def build(force=False):
    src = os.path.join(HERE, 'nvdr_oracle.c')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-B', '_build/libnvdr_oracle.so'], stdout=subprocess.DEVNULL)
    return LIB


def build_ref(force=False):
    """Build oracle/_ref from the reference checkout (no-op when it is absent, e.g. on the GPU box)."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, 'render', 'optixutils', 'c_src')):
        return None
    if force or not (os.path.exists(REF_LIB) and os.path.exists(REF_DM_LIB)):
        subprocess.check_call(['make', '-C', HERE, '-B', 'ref', 'REF=' + REFERENCE_ROOT], stdout=subprocess.DEVNULL)
    return REF_LIB


_libs = {}


def _load(path):
    if path not in _libs:
        if path == LIB:
            build()
        _libs[path] = ctypes.CDLL(path)
    return _libs[path]


def have_ref():
    return os.path.exists(REF_LIB) and os.path.exists(REF_DM_LIB)


def max_threads():
    lib = _load(LIB)
    return int(lib.oracle_max_threads())


def _cpu(t, dtype=torch.float32):
    assert isinstance(t, torch.Tensor) and t.device.type == 'cpu' and t.dtype == dtype, (type(t), getattr(t, 'device', None), getattr(t, 'dtype', None))
    return t


def _view(t, ndim):
    """Tensor view padded with TRAILING size-1 dims (the C side indexes dims 0..ndim-1)."""
    v = Tensor()
    sizes, strides = list(t.shape), list(t.stride())
    assert len(sizes) == ndim, (sizes, ndim)
    sizes += [1] * (4 - ndim)
    strides += [0] * (4 - ndim)
    v.data = t.data_ptr()
    for i in range(4):
        v.size[i] = sizes[i]
        v.stride[i] = strides[i]
    return v


def _shade_args(mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf,
                n_samples_x, rnd_seed, shadow_scale, pixel_index_offset):
    a = EnvShadeArgs()
    a.mask = _view(_cpu(mask), 3)
    for name, t in (('ro', ro), ('gb_pos', gb_pos), ('gb_normal', gb_normal), ('gb_view_pos', gb_view_pos),
                    ('gb_kd', gb_kd), ('gb_ks', gb_ks)):
        setattr(a, name, _view(_cpu(t), 4))
    a.light = _view(_cpu(light), 3)
    a.pdf = _view(_cpu(pdf), 2)
    a.rows = _view(_cpu(rows), 1)
    a.cols = _view(_cpu(cols), 2)
    a.perms = _view(_cpu(perms, torch.int32), 2)
    a.bsdf = ['pbr', 'diffuse', 'white'].index(bsdf) if isinstance(bsdf, str) else int(bsdf)
    a.n_samples_x = n_samples_x
    a.rnd_seed = rnd_seed
    a.shadow_scale = shadow_scale
    a.pixel_index_offset = pixel_index_offset
    return a


def env_shade(verts, tris, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
              bsdf='pbr', n_samples_x=8, rnd_seed=0, shadow_scale=1.0, diff_grad=None, spec_grad=None,
              n_threads=1, vis_in=None, want_vis=False, want_dbg=False, pixel_index_offset=0, impl='oracle', frozen=None):
    """Forward (diff_grad is None) or backward pass.  impl: 'oracle' | 'ref' | 'ref_detmath'.
    Returns a dict: fwd {diff, spec}, bwd {gb_pos_grad, gb_normal_grad, gb_kd_grad, gb_ks_grad, light_grad};
    plus 'covered', optionally 'vis' [N*H*W, 2S] uint8 and 'dbg' [N*H*W, 2S, 4].
    frozen (impl 'oracle' only): dict with any of gb_pos / gb_normal / gb_view_pos / gb_kd / gb_ks -- the G-buffer the SAMPLES
    are generated from (directions, pdfs, lobe choice), while the arguments proper are what is evaluated at the samples."""
    a = _shade_args(mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf,
                    n_samples_x, rnd_seed, shadow_scale, pixel_index_offset)
    N, H, W = ro.shape[0], ro.shape[1], ro.shape[2]
    S = n_samples_x * n_samples_x
    backward = diff_grad is not None
    out = {}
    keep = []
    if not backward:
        out['diff'] = torch.empty(N, H, W, 3)
        out['spec'] = torch.empty(N, H, W, 3)
        a.diff, a.spec = out['diff'].data_ptr(), out['spec'].data_ptr()
    else:
        a.diff_grad = _view(_cpu(diff_grad), 4)
        a.spec_grad = _view(_cpu(spec_grad), 4)
        for k in ('gb_pos_grad', 'gb_normal_grad', 'gb_kd_grad', 'gb_ks_grad'):
            out[k] = torch.empty(N, H, W, 3)
            setattr(a, k, out[k].data_ptr())
        out['light_grad'] = torch.empty(light.shape[0], light.shape[1], 3)
        a.light_grad = out['light_grad'].data_ptr()
    verts = _cpu(verts).contiguous()
    tris = _cpu(tris, torch.int32).contiguous()
    keep += [verts, tris]
    vin = None
    if vis_in is not None:
        vin = vis_in.contiguous()
        assert vin.dtype == torch.uint8 and vin.numel() == N * H * W * 2 * S
    vout = torch.zeros(N * H * W, 2 * S, dtype=torch.uint8) if want_vis else None
    dbg = torch.zeros(N * H * W, 2 * S, 4) if want_dbg else None
    P = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
    if impl == 'oracle':
        lib = _load(LIB)
        lib.oracle_env_shade_frozen.restype = c_long
        sa = a
        if frozen:
            g = dict(gb_pos=gb_pos, gb_normal=gb_normal, gb_view_pos=gb_view_pos, gb_kd=gb_kd, gb_ks=gb_ks)
            g.update(frozen)
            sa = _shade_args(mask, ro, g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'], light, pdf, rows, cols,
                             perms, bsdf, n_samples_x, rnd_seed, shadow_scale, pixel_index_offset)
            keep += list(g.values())
        cov = lib.oracle_env_shade_frozen(ctypes.byref(a), ctypes.byref(sa), P(verts), P(tris), c_long(tris.shape[0]), c_int(int(backward)),
                                          c_int(n_threads), P(vin), P(vout), P(dbg))
    else:
        assert not want_dbg and not frozen
        lib = _load(REF_LIB if impl == 'ref' else REF_DM_LIB)
        lib.ref_env_shade.restype = c_long
        cov = lib.ref_env_shade(ctypes.byref(a), P(verts), P(tris), c_long(tris.shape[0]), c_int(int(backward)),
                                c_int(n_threads), P(vin), P(vout))
    out['covered'] = int(cov)
    if want_vis:
        out['vis'] = vout
    if want_dbg:
        out['dbg'] = dbg
    return out


def bilateral_fwd(col, nrm, zdz, sigma, n_threads=1, impl='oracle'):
    """-> [N,H,W,4] (sum rgb*w, max(sum w, 1e-4)), as bilateral_denoiser_fwd (torch_bindings.cpp:274-295)."""
    N, H, W = col.shape[:3]
    out = torch.empty(N, H, W, 4)
    vc, vn, vz = _view(_cpu(col), 4), _view(_cpu(nrm), 4), _view(_cpu(zdz), 4)
    if impl == 'oracle':
        _load(LIB).oracle_bilateral_fwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), c_float(sigma),
                                        c_void_p(out.data_ptr()), c_int(n_threads))
    else:
        _load(REF_LIB).ref_bilateral_fwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), c_float(sigma),
                                         c_void_p(out.data_ptr()), c_int(n_threads))
    return out


def bilateral_bwd(col, nrm, zdz, sigma, out_grad, n_threads=1, impl='oracle'):
    N, H, W = col.shape[:3]
    g = torch.empty(N, H, W, 3)
    vc, vn, vz, vg = _view(_cpu(col), 4), _view(_cpu(nrm), 4), _view(_cpu(zdz), 4), _view(_cpu(out_grad), 4)
    if impl == 'oracle':
        _load(LIB).oracle_bilateral_bwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), c_float(sigma),
                                        ctypes.byref(vg), c_void_p(g.data_ptr()), c_int(n_threads))
    else:
        _load(REF_LIB).ref_bilateral_bwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), c_float(sigma),
                                         ctypes.byref(vg), c_void_p(g.data_ptr()), c_int(n_threads))
    return g


def bilateral_denoiser(col, nrm, zdz, sigma, n_threads=1, impl='oracle'):
    """The full op of render/optixutils/ops.py:139-141: weighted sum divided by the weight sum."""
    o = bilateral_fwd(col, nrm, zdz, sigma, n_threads, impl)
    return o[..., 0:3] / o[..., 3:4]


def visibility(verts, tris, ro, rd, n_threads=1):
    """uint8 [R]: 1 = the ray hits nothing in (0, 1e16)."""
    verts, tris = _cpu(verts).contiguous(), _cpu(tris, torch.int32).contiguous()
    ro, rd = _cpu(ro).contiguous(), _cpu(rd).contiguous()
    R = ro.shape[0]
    out = torch.empty(R, dtype=torch.uint8)
    _load(LIB).oracle_visibility(c_void_p(verts.data_ptr()), c_void_p(tris.data_ptr()), c_long(tris.shape[0]),
                                 c_void_p(ro.data_ptr()), c_void_p(rd.data_ptr()), c_long(R), c_void_p(out.data_ptr()),
                                 c_int(n_threads))
    return out


def closest(verts, tris, ro, rd, n_threads=1):
    verts, tris = _cpu(verts).contiguous(), _cpu(tris, torch.int32).contiguous()
    ro, rd = _cpu(ro).contiguous(), _cpu(rd).contiguous()
    R = ro.shape[0]
    t = torch.empty(R)
    tri = torch.empty(R, dtype=torch.int32)
    uv = torch.empty(R, 2)
    _load(LIB).oracle_closest(c_void_p(verts.data_ptr()), c_void_p(tris.data_ptr()), c_long(tris.shape[0]),
                              c_void_p(ro.data_ptr()), c_void_p(rd.data_ptr()), c_long(R), c_void_p(t.data_ptr()),
                              c_void_p(tri.data_ptr()), c_void_p(uv.data_ptr()), c_int(n_threads))
    return t, tri, uv


def bvh2_walk(nodes, trirec, grid_lo, grid_scale, ro, rd, n_threads=1):
    """CPU walk of the binary tree exported by the GPU (nvdr_bvh_export): -> (vis uint8 [R], node visits, triangle tests).
    nodes: numpy uint32 [n_nodes, 8]; trirec: numpy float32 [n_tris, 12]; grid_lo / grid_scale: 3 floats each."""
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    trirec = np.ascontiguousarray(trirec, dtype=np.float32)
    ro, rd = _cpu(ro).contiguous(), _cpu(rd).contiguous()
    R = ro.shape[0]
    out = torch.empty(R, dtype=torch.uint8)
    lo = (c_float * 3)(*grid_lo)
    sc = (c_float * 3)(*grid_scale)
    cnt = (ctypes.c_longlong * 2)(0, 0)
    _load(LIB).oracle_bvh2_walk(nodes.ctypes.data_as(c_void_p), trirec.ctypes.data_as(c_void_p), c_long(trirec.shape[0]), lo, sc,
                                c_void_p(ro.data_ptr()), c_void_p(rd.data_ptr()), c_long(R), c_void_p(out.data_ptr()), cnt,
                                c_int(n_threads))
    return out, int(cnt[0]), int(cnt[1])


def light_update_pdf(base):
    base = _cpu(base).contiguous()
    H, W = base.shape[0], base.shape[1]
    pdf, cols, rows = torch.empty(H, W), torch.empty(H, W), torch.empty(H)
    _load(LIB).oracle_light_update_pdf(c_void_p(base.data_ptr()), c_long(H), c_long(W), c_void_p(pdf.data_ptr()),
                                       c_void_p(cols.data_ptr()), c_void_p(rows.data_ptr()))
    return pdf, cols, rows


def detmath(op, x, y=None):
    """include/nvdr_detmath.h on the host: op in ('sin', 'cos', 'acos', 'atan2'); atan2 computes atan2(x, y)."""
    x = _cpu(x).contiguous()
    y = x if y is None else _cpu(y).contiguous()
    out = torch.empty_like(x)
    _load(LIB).oracle_detmath(c_int(['sin', 'cos', 'acos', 'atan2'].index(op)), c_void_p(x.data_ptr()), c_void_p(y.data_ptr()),
                              c_long(x.numel()), c_void_p(out.data_ptr()))
    return out
