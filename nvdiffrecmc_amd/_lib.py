"""ctypes binding of libnvdr_hip.so (C ABI: include/nvdr_hip.h).

This is the binding a maintainer of the reference would write in place of the two
torch.utils.cpp_extension.load() calls (render/optixutils/ops.py:67-72,
render/renderutils/ops.py:78-83).  There is NO fallback: if the HIP library is missing or a
call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: maps the HIP runtime the library binds to)

from . import _build

c_void_p, c_int, c_int64, c_float, c_uint32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32


class NvdrTensor(ctypes.Structure):
    _fields_ = [('data', c_void_p), ('size', c_int64 * 4), ('stride', c_int64 * 4)]


class NvdrBvhInfo(ctypes.Structure):
    _fields_ = [('n_tris', c_int64), ('n_nodes', c_int64), ('height', ctypes.c_int32), ('root', ctypes.c_int32),
                ('aabb_min', c_float * 3), ('aabb_max', c_float * 3), ('grid_lo', c_float * 3), ('grid_scale', c_float * 3),
                ('stack_max', ctypes.c_int32)]

COUNTERS_BVH2 = 8 + 2 * 8192   # NVDR_COUNTERS_BVH2
COUNTERS_PHASES = COUNTERS_BVH2 + 8  # NVDR_COUNTERS_PHASES
COUNTERS_LEN = COUNTERS_PHASES + 16


class NvdrEnvShadeArgs(ctypes.Structure):
    _fields_ = [(n, NvdrTensor) for n in ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks',
                                          'light', 'pdf', 'rows', 'cols', 'perms')] + [
        ('bsdf', c_uint32), ('n_samples_x', c_uint32), ('rnd_seed', c_uint32), ('shadow_scale', c_float),
        ('pixel_index_offset', c_uint32),
        ('diff', c_void_p), ('spec', c_void_p),
        ('diff_grad', NvdrTensor), ('spec_grad', NvdrTensor),
        ('gb_pos_grad', c_void_p), ('gb_normal_grad', c_void_p), ('gb_kd_grad', c_void_p), ('gb_ks_grad', c_void_p),
        ('light_grad', c_void_p), ('vis_cache', c_void_p), ('counters', c_void_p), ('reuse_stream_id', ctypes.c_uint64), ('rnd_seed_offset', c_void_p),
        ('rnd_seed_snapshot', c_void_p), ('rnd_seed_advance', ctypes.c_uint32), ('phase', ctypes.c_uint32)]


class NvdrGbufferArgs(ctypes.Structure):
    _fields_ = [('v_pos', c_void_p), ('t_pos_idx', c_void_p), ('v_nrm', c_void_p), ('t_nrm_idx', c_void_p),
                ('v_tng', c_void_p), ('t_tng_idx', c_void_p), ('v_tex', c_void_p), ('t_tex_idx', c_void_p),
                ('n_tris', c_int64), ('mvp', c_void_p), ('cam', c_void_p),
                ('n', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32)] + [
        (k, c_void_p) for k in ('rast', 'rast_db', 'gb_pos', 'gb_geometric_normal', 'gb_normal', 'gb_tangent', 'gb_texc',
                                'gb_texc_deriv', 'gb_depth')]


class NvdrMeshArgs(ctypes.Structure):          # include/nvdr_hip.h: nvdr_mesh_args
    _fields_ = [('v_pos', c_void_p), ('n_verts', c_int64), ('t_pos_idx', c_void_p), ('n_tris', c_int64),
                ('v_tex', c_void_p), ('t_tex_idx', c_void_p), ('adj_start', c_void_p), ('adj_corner', c_void_p)]


class NvdrInterpolateBwdArgs(ctypes.Structure):   # nvdr_interpolate_bwd_args
    _fields_ = [('rast', c_void_p), ('n', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('v_pos', c_void_p), ('t_pos_idx', c_void_p), ('v_nrm', c_void_p), ('v_tng', c_void_p),
                ('n_verts', c_int64), ('n_tris', c_int64), ('cam', c_void_p),
                ('gb_pos_grad', c_void_p), ('gb_geometric_normal_grad', c_void_p), ('gb_normal_grad', c_void_p), ('gb_tangent_grad', c_void_p),
                ('v_pos_grad', c_void_p), ('v_nrm_grad', c_void_p), ('v_tng_grad', c_void_p)]


MAX_TEXTURES = 4


class NvdrTextureArgs(ctypes.Structure):        # nvdr_texture_args
    _fields_ = [('n_tex', ctypes.c_int32), ('res', ctypes.c_int32 * MAX_TEXTURES), ('tex', c_void_p * MAX_TEXTURES),
                ('texc', c_void_p), ('rast', c_void_p), ('n_pix', c_int64), ('out', c_void_p * MAX_TEXTURES),
                ('dout', c_void_p * MAX_TEXTURES), ('dtex', c_void_p * MAX_TEXTURES), ('accumulate', ctypes.c_int32)]


ALLOC_FN = ctypes.CFUNCTYPE(c_void_p, ctypes.c_size_t, c_int, c_void_p, c_void_p)     # nvdr_alloc_fn
FREE_FN = ctypes.CFUNCTYPE(None, c_void_p, c_void_p)                                    # nvdr_free_fn


def torch_allocator():
    """(alloc, free) callbacks that serve the library's scratch from torch's caching allocator (nvdr_ctx_set_allocator)."""
    def _alloc(nbytes, device, stream, user):
        try:
            return torch.cuda.caching_allocator_alloc(int(nbytes), int(device), int(stream or 0))
        except Exception:           # out of memory etc.: NULL -> the library reports the failed allocation
            return 0

    def _free(ptr, user):
        try:
            torch.cuda.caching_allocator_delete(int(ptr))
        except Exception:           # interpreter shutdown: the process is going away anyway
            pass
    return ALLOC_FN(_alloc), FREE_FN(_free)


_T = ctypes.POINTER(NvdrTensor)


class NvdrAdamTensor(ctypes.Structure):     # include/nvdr_hip.h: nvdr_adam_tensor
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('exp_avg', c_void_p), ('exp_avg_sq', c_void_p), ('n', c_int64),
                ('grad_scale', c_float), ('lo', c_float), ('hi', c_float), ('lo_vec', c_void_p), ('lo_vec_n', c_int64),
                ('hi_vec', c_void_p), ('hi_vec_n', c_int64), ('lr_scale', c_float), ('normalize3', ctypes.c_int32),
                ('active', c_void_p), ('zero_grad', ctypes.c_int32), ('frozen', ctypes.c_int32)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    'nvdr_last_error': [],
    'nvdr_version': [],
    'nvdr_ctx_create': [ctypes.POINTER(c_void_p), c_int],
    'nvdr_ctx_destroy': [c_void_p],
    'nvdr_ctx_check': [c_void_p, c_void_p],
    'nvdr_ctx_set_stream_budget': [c_void_p, c_int64],
    'nvdr_ctx_set_allocator': [c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_ctx_set_build_mode': [c_void_p, c_int],
    'nvdr_bvh_wait': [c_void_p, c_void_p],
    'nvdr_bvh_mark_joined': [c_void_p],
    'nvdr_bvh_build': [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p],
    'nvdr_bvh_info_get': [c_void_p, ctypes.POINTER(NvdrBvhInfo), c_void_p],
    'nvdr_bvh_export': [c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_bvh_export_oct': [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p],
    'nvdr_trace_visibility': [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    'nvdr_trace_visibility_wide': [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    'nvdr_trace_visibility_wide_counted': [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    'nvdr_trace_closest': [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_render_gbuffer': [c_void_p, ctypes.POINTER(NvdrGbufferArgs), c_void_p],
    'nvdr_mesh_frame_fwd': [ctypes.POINTER(NvdrMeshArgs), c_void_p, c_void_p, c_void_p],
    'nvdr_mesh_frame_bwd': [ctypes.POINTER(NvdrMeshArgs), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    'nvdr_interpolate_bwd': [ctypes.POINTER(NvdrInterpolateBwdArgs), c_void_p],
    'nvdr_texture_lookup_fwd': [ctypes.POINTER(NvdrTextureArgs), c_void_p],
    'nvdr_texture_lookup_bwd': [ctypes.POINTER(NvdrTextureArgs), c_void_p],
    'nvdr_env_shade_fwd': [c_void_p, ctypes.POINTER(NvdrEnvShadeArgs), c_void_p],
    'nvdr_env_shade_bwd': [c_void_p, ctypes.POINTER(NvdrEnvShadeArgs), c_void_p],
    'nvdr_env_shade_last_pixel_count': [c_void_p, ctypes.POINTER(c_int64), c_void_p],
    'nvdr_env_shade_stream_id': [c_void_p, ctypes.POINTER(ctypes.c_uint64)],
    'nvdr_ctx_set_profiling': [c_void_p, c_int],
    'nvdr_env_shade_stage_times': [c_void_p, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)],
    'nvdr_bilateral_denoiser_fwd': [_T, _T, _T, c_float, c_void_p, c_void_p],
    'nvdr_bilateral_denoiser_bwd': [_T, _T, _T, c_float, _T, c_void_p, c_void_p],
    'nvdr_bilateral_denoiser_pair_fwd': [_T, _T, _T, _T, c_float, c_void_p, c_void_p, c_void_p],
    'nvdr_bilateral_denoiser_pair_bwd': [_T, _T, _T, c_float, _T, _T, c_void_p, c_void_p, c_void_p],
    'nvdr_image_loss_num_partials': [c_int64, c_int64, c_int64],
    'nvdr_image_loss_fwd': [_T, _T, c_int, c_int, c_void_p, c_void_p],
    'nvdr_image_loss_bwd': [_T, _T, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_image_loss_mean_fwd': [_T, _T, c_int, c_int, c_void_p, c_void_p, c_void_p],
    'nvdr_image_loss_mean_bwd': [_T, _T, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_prepare_shading_normal_fwd': [_T] * 6 + [c_int, c_int, c_void_p, c_void_p],
    'nvdr_shading_frame_fwd': [_T] * 6 + [c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_prepare_shading_normal_bwd': [_T] * 7 + [c_int, c_int] + [c_void_p] * 6 + [c_void_p],
    'nvdr_xfm_fwd': [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    'nvdr_xfm_bwd': [c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p],
    'nvdr_lambert_fwd': [_T, _T, c_void_p, c_void_p],
    'nvdr_lambert_bwd': [_T, _T, _T, c_void_p, c_void_p, c_void_p],
    'nvdr_frostbite_fwd': [_T] * 4 + [c_void_p, c_void_p],
    'nvdr_frostbite_bwd': [_T] * 5 + [c_void_p] * 4 + [c_void_p],
    'nvdr_fresnel_shlick_fwd': [_T] * 3 + [c_void_p, c_void_p],
    'nvdr_fresnel_shlick_bwd': [_T] * 4 + [c_void_p] * 3 + [c_void_p],
    'nvdr_ndf_ggx_fwd': [_T] * 2 + [c_void_p, c_void_p],
    'nvdr_ndf_ggx_bwd': [_T] * 3 + [c_void_p] * 2 + [c_void_p],
    'nvdr_lambda_ggx_fwd': [_T] * 2 + [c_void_p, c_void_p],
    'nvdr_lambda_ggx_bwd': [_T] * 3 + [c_void_p] * 2 + [c_void_p],
    'nvdr_masking_smith_fwd': [_T] * 3 + [c_void_p, c_void_p],
    'nvdr_masking_smith_bwd': [_T] * 4 + [c_void_p] * 3 + [c_void_p],
    'nvdr_pbr_specular_fwd': [_T] * 5 + [c_float, c_void_p, c_void_p],
    'nvdr_pbr_specular_bwd': [_T] * 5 + [c_float, _T] + [c_void_p] * 5 + [c_void_p],
    'nvdr_pbr_bsdf_fwd': [_T] * 6 + [c_float, c_int, c_void_p, c_void_p],
    'nvdr_pbr_bsdf_bwd': [_T] * 6 + [c_float, c_int, _T] + [c_void_p] * 6 + [c_void_p],
    'nvdr_shade_composite_fwd': [_T] * 4 + [c_int, c_void_p, c_void_p],
    'nvdr_shade_composite_bwd': [_T] * 4 + [c_int, _T] + [c_void_p] * 4 + [c_void_p],
    'nvdr_shade_loss_fused': [_T] * 4 + [c_int, _T, c_int, c_int] + [c_void_p] * 7 + [c_void_p],
    'nvdr_gather_rows_fwd': [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    'nvdr_gather_rows_bwd': [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p],
    'nvdr_light_update_pdf': [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    'nvdr_adam_step': [ctypes.POINTER(NvdrAdamTensor), c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p],
    'nvdr_adam_step_partial': [ctypes.POINTER(NvdrAdamTensor), c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p, c_int, c_void_p],
    'nvdr_abi_sizeof': [c_int],
    'nvdr_tile_flags': [c_void_p, c_int64, c_int, c_void_p, c_void_p],
    'nvdr_tile_plan': [c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    'nvdr_tile_gather': [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    'nvdr_tile_scatter': [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    'nvdr_test_detmath': [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    'nvdr_test_arith': [c_void_p, c_void_p],
}
_RESTYPES = {'nvdr_last_error': ctypes.c_char_p, 'nvdr_image_loss_num_partials': c_int64, 'nvdr_abi_sizeof': ctypes.c_size_t}
# nvdr_abi_sizeof(which) -> the ctypes mirror that must have that size (checked once at load: a stale library or a stale mirror is an error)
_ABI_MIRRORS = {0: 'NvdrAdamTensor', 1: 'NvdrEnvShadeArgs', 2: 'NvdrTextureArgs', 3: 'NvdrInterpolateBwdArgs', 4: 'NvdrTensor', 5: 'NvdrGbufferArgs',
                6: 'NvdrMeshArgs', 7: 'NvdrBvhInfo'}

EXPORTED_SYMBOLS = sorted(_SIGNATURES)

_lib = None


def tuning_env(name, default=None):
    """Value of an experiment switch (NVDR_RAW_ALLOC, NVDR_PAIR_FILTER, ...): read only when NVDR_TUNING=1 is set as well, like the
    library's own switches (csrc/core.hip nvdr_tuning_env); a stray variable in a production environment changes nothing."""
    if os.environ.get('NVDR_TUNING', '0') not in ('', '0'):
        return os.environ.get(name, default)
    return default


def lib_path():
    return _build.LIB


def load():
    """Load libnvdr_hip.so.  Raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libnvdr_hip.so is not built (%s). Build the HIP extension first: "
            "python -c 'import __graft_entry__ as g; g.build()'  -- there is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the library is stale: rebuild
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    for which, mirror in _ABI_MIRRORS.items():
        want, have = int(lib.nvdr_abi_sizeof(which)), ctypes.sizeof(globals()[mirror])
        if want != have:
            raise RuntimeError('libnvdr_hip.so and its ctypes mirror disagree on %s (%d vs %d bytes): rebuild the library '
                               '(__graft_entry__.build()) or update nvdiffrecmc_amd/_lib.py' % (mirror, want, have))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().nvdr_last_error()
        raise RuntimeError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else '?'))


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_cuda_f32(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must reside on the GPU (got %s); this path has no CPU fallback' % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError('%s must have dtype %s (got %s)' % (name, dtype, t.dtype))


def tensor_view(t, ndim_to=4, lead=True):
    """nvdr_tensor for `t`, padded to 4 dims with size-1/stride-0 dims (leading by default)."""
    v = NvdrTensor()
    if t is None:
        v.data = None
        for i in range(4):
            v.size[i] = 1
            v.stride[i] = 0
        return v
    sizes, strides = list(t.shape), list(t.stride())
    assert len(sizes) <= 4
    pad = 4 - len(sizes)
    if lead:
        sizes, strides = [1] * pad + sizes, [0] * pad + strides
    else:
        sizes, strides = sizes + [1] * pad, strides + [0] * pad
    v.data = t.data_ptr()
    for i in range(4):
        v.size[i] = sizes[i]
        v.stride[i] = strides[i]
    return v
