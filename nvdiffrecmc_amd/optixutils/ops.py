"""Drop-in replacement for render/optixutils/ops.py on MI355X (PyTorch-ROCm + libnvdr_hip.so).

Same names, argument order and autograd contract as the reference module:

    OptiXContext()                                                    ops.py:125-128
    optix_build_bvh(optix_ctx, verts, tris, rebuild)                  ops.py:130-133
    optix_env_shade(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks,
                    light, pdf, rows, cols, BSDF='pbr', n_samples_x=8,
                    rnd_seed=None, shadow_scale=1.0) -> (diff, spec)  ops.py:135-137
    bilateral_denoiser(col, nrm, zdz, sigma) -> [N,H,W,3]             ops.py:139-141

There is no OptiX, no NVRTC and no CUDA underneath: the context owns an LBVH in HBM and the
kernels are hand-written HIP for gfx950 (nvdiffrecmc_amd/csrc).  There is no CPU fallback either:
tensors must live on the GPU and the HIP library must be built, otherwise RuntimeError.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib

__all__ = ["OptiXContext", "optix_build_bvh", "optix_env_shade", "bilateral_denoiser"]


class _HipContext:
    """Owns the nvdr_ctx handle (the role of OptiXStateWrapper, optix_wrapper.h:17-37)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("OptiXContext needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = ctypes.c_void_p()
        _lib.check(self.lib.nvdr_ctx_create(ctypes.byref(h), self.device), 'nvdr_ctx_create')
        self.handle = h
        # scratch (ray stream, stack spill, BVH buffers) comes from torch's caching allocator: visible in torch.cuda.memory_*
        # and returned to its pool when the context dies.  NVDR_RAW_ALLOC=1 keeps the library on hipMalloc (A/B, debugging).
        self._alloc_cb = None
        if not _lib.tuning_env('NVDR_RAW_ALLOC'):
            self._alloc_cb = _lib.torch_allocator()
            _lib.check(self.lib.nvdr_ctx_set_allocator(h, ctypes.cast(self._alloc_cb[0], ctypes.c_void_p),
                                                       ctypes.cast(self._alloc_cb[1], ctypes.c_void_p), None), 'nvdr_ctx_set_allocator')
        self._geom = None  # keeps the verts/tris tensors alive while the device may still read them

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.nvdr_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class OptiXContext:
    """API-compatible stand-in for the reference's OptiXContext (ops.py:125-128): attribute
    `.cpp_wrapper` holds the native state, as in the reference."""

    def __init__(self, device=None):
        self.cpp_wrapper = _HipContext(device)
        # Behavioural switches live on the CONTEXT (round 1 kept them in process-global class attributes):
        #   cache_visibility   reuse the forward's visibility bits in backward when the seed is fixed (identical rays; the
        #                      reference re-traces every ray, torch_bindings.cpp:238,266).  None = the module default.
        #   pixel_index_offset added to the linear pixel index that seeds the RNG (data-parallel shards: first_view * H * W)
        #   seed_offset        int32 device tensor [1] the kernels add to rnd_seed (None: no offset).  A captured HIP graph
        #                      freezes host-side launch parameters; a seed counter in device memory lets replays draw new samples
        self.cache_visibility = None
        self.pixel_index_offset = None
        self.seed_offset = None
        self.seed_advance = 0       # != 0 (with seed_offset and a fixed rnd_seed): every forward launch adds this to the device counter itself
        #   split_hook         callable or None: the forward launch is issued in two calls (nvdr_env_shade_args.phase) and the hook runs between
        #                      them, i.e. between the sample generation and the first kernel that reads the BVH (trainer.py ends one HIP
        #                      graph and begins the next there)
        self.split_hook = None

    def set_stream_budget(self, megabytes):
        """HBM the ray stream between the three env-shade stages may take (default 8192 MB); larger launches are processed
        in chunks of covered pixels with identical results."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_ctx_set_stream_budget(w.handle, int(megabytes) << 20), 'nvdr_ctx_set_stream_budget')

    def wait_build(self):
        """The current stream waits (on the device) for the last optix_build_bvh of this context; consumers of the tree do this by themselves."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_bvh_wait(w.handle, _lib.stream_ptr()), 'nvdr_bvh_wait')

    def build_joined(self):
        """The caller orders its streams behind the last build itself (nvdr_bvh_mark_joined): consumers issue no wait of their own."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_bvh_mark_joined(w.handle), 'nvdr_bvh_mark_joined')

    def set_build_mode(self, mode):
        """Where optix_build_bvh runs: 1 (default) side stream, 0 the caller's stream, 2 side stream with the launches deferred to the
        first consumer of the tree (HIP graphs of launch-bound iterations: the caller's own front nodes come first)."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_ctx_set_build_mode(w.handle, int(mode)), 'nvdr_ctx_set_build_mode')

    def check(self):
        """Synchronise and raise if any traversal launch on this context ever overflowed its stack (never silent)."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_ctx_check(w.handle, _lib.stream_ptr()), 'nvdr_ctx_check')

    def set_profiling(self, enable=True):
        """Record HIP events around the three env-shade stages of every launch on this context (bench.py)."""
        w = self.cpp_wrapper
        _lib.check(w.lib.nvdr_ctx_set_profiling(w.handle, int(bool(enable))), 'nvdr_ctx_set_profiling')

    def stage_times(self, backward=False):
        """(launches, [gen_ms, trace_ms, shade_ms] averaged) over the recorded forward or backward launches."""
        w = self.cpp_wrapper
        ms = (ctypes.c_double * 3)()
        n = ctypes.c_int64()
        _lib.check(w.lib.nvdr_env_shade_stage_times(w.handle, int(bool(backward)), ms, ctypes.byref(n)), 'stage_times')
        k = max(int(n.value), 1)
        return int(n.value), [ms[0] / k, ms[1] / k, ms[2] / k]

    def bvh_info(self):
        w = self.cpp_wrapper
        info = _lib.NvdrBvhInfo()
        _lib.check(w.lib.nvdr_bvh_info_get(w.handle, ctypes.byref(info), _lib.stream_ptr()), 'nvdr_bvh_info_get')
        return {'n_tris': info.n_tris, 'n_nodes': info.n_nodes, 'height': info.height,
                'aabb_min': list(info.aabb_min), 'aabb_max': list(info.aabb_max),
                'grid_lo': list(info.grid_lo), 'grid_scale': list(info.grid_scale), 'stack_max': info.stack_max}

    def bvh_export(self):
        """Host copies of the binary tree: nodes uint32 [n_nodes, 8] (32-B records, csrc/bvh.h) and triangle records
        float32 [n_tris, 12] in Morton order (v0, e1, e2, original index bits, 0, 0)."""
        w = self.cpp_wrapper
        info = self.bvh_info()
        nodes = np.zeros((max(info['n_nodes'], 1), 8), dtype=np.uint32)
        tris = np.zeros((info['n_tris'], 12), dtype=np.float32)
        _lib.check(w.lib.nvdr_bvh_export(w.handle, nodes.ctypes.data_as(ctypes.c_void_p), tris.ctypes.data_as(ctypes.c_void_p),
                                         _lib.stream_ptr()), 'nvdr_bvh_export')
        return nodes[:info['n_nodes']], tris


def _bvh_export_oct(self):
    """Host copies of the eight-wide tree the shadow rays walk: oct uint32 [n_oct, 16] (64-B records, csrc/bvh.h) and the
    triangle records float32 [n_tris, 12] in the order the oct nodes refer to them; plus the builder's counters."""
    w = self.cpp_wrapper
    cnt = (ctypes.c_int64 * 3)()
    _lib.check(w.lib.nvdr_bvh_export_oct(w.handle, None, None, cnt, _lib.stream_ptr()), 'nvdr_bvh_export_oct')
    n_oct, n_tri = int(cnt[0]), int(cnt[1])
    oct = np.zeros((max(n_oct, 1), 16), dtype=np.uint32)
    tris8 = np.zeros((self.bvh_info()['n_tris'], 12), dtype=np.float32)
    _lib.check(w.lib.nvdr_bvh_export_oct(w.handle, oct.ctypes.data_as(ctypes.c_void_p), tris8.ctypes.data_as(ctypes.c_void_p), cnt,
                                         _lib.stream_ptr()), 'nvdr_bvh_export_oct')
    return oct[:n_oct], tris8, {'nodes': n_oct, 'triangles_placed': n_tri, 'nodes_finished': int(cnt[2])}


OptiXContext.bvh_export_oct = _bvh_export_oct


def optix_build_bvh(optix_ctx, verts, tris, rebuild):
    """Build (rebuild > 0) or refit (rebuild == 0) the BVH held by `optix_ctx` (ops.py:130-133)."""
    assert tris.shape[0] > 0, "Got empty training triangle mesh (unrecoverable discontinuity)"
    assert verts.shape[0] > 0, "Got empty training triangle mesh (unrecoverable discontinuity)"
    w = optix_ctx.cpp_wrapper
    verts = verts.detach().view(-1, 3)
    tris = tris.detach().view(-1, 3)
    _lib.require_cuda_f32(verts, 'verts')
    _lib.require_cuda_f32(tris, 'tris', torch.int32)
    verts, tris = verts.contiguous(), tris.contiguous()
    w._geom = (verts, tris)
    _lib.check(w.lib.nvdr_bvh_build(w.handle, _lib.ptr(verts), verts.shape[0], _lib.ptr(tris), tris.shape[0],
                                     int(rebuild), _lib.stream_ptr()), 'optix_build_bvh')


# ----------------------------------------------------------------------------------------------

def _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
               BSDF, n_samples_x, rnd_seed, shadow_scale, pixel_index_offset, seed_offset=None):
    a = _lib.NvdrEnvShadeArgs()
    for name, t, nd in (('mask', mask, 3), ('ro', ro, 4), ('gb_pos', gb_pos, 4), ('gb_normal', gb_normal, 4),
                        ('gb_view_pos', gb_view_pos, 4), ('gb_kd', gb_kd, 4), ('gb_ks', gb_ks, 4), ('light', light, 3),
                        ('pdf', pdf, 2), ('rows', rows, 1), ('cols', cols, 2)):
        _lib.require_cuda_f32(t, name)
        if t.dim() != nd:
            raise RuntimeError('%s must have %d dims (got shape %s)' % (name, nd, tuple(t.shape)))
        setattr(a, name, _lib.tensor_view(t, lead=False))
    _lib.require_cuda_f32(perms, 'perms', torch.int32)
    a.perms = _lib.tensor_view(perms, lead=False)
    a.bsdf, a.n_samples_x, a.rnd_seed = int(BSDF), int(n_samples_x), int(rnd_seed) & 0xFFFFFFFF
    a.shadow_scale = float(shadow_scale)
    a.pixel_index_offset = int(pixel_index_offset)
    if seed_offset is not None:
        _lib.require_cuda_f32(seed_offset, 'seed_offset', torch.int32)
        a.rnd_seed_offset = seed_offset.data_ptr()
    return a


PERM_TABLE_SEED = 0x5EED      # seed of the default permutation tables (_perms_for)


def _perms_for(n_samples_x, device):
    """(32k) table of random permutations that decorrelate the BSDF and light strata: one per (device, n_samples_x),
    created lazily and cached for the process (ops.py:79,84-86; the reference keys by n_samples_x only -- it runs one
    device per process).  Drawn from a generator of its own with a FIXED seed (the reference draws from the global CUDA generator):
    the ranks of a data-parallel run are separate processes, and a view rendered on rank r equals the same view inside the one-GPU
    batch launch only if both stratify with the same table -- and so does every run.  PERM_TABLE_SEED / set_permutation_table()
    choose another table; the parity tests inject a CPU-generated one."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), n_samples_x)
    t = _optix_env_shade_func._random_perm.get(key)
    if t is None:
        g = torch.Generator(device=device)
        g.manual_seed(PERM_TABLE_SEED + n_samples_x)
        t = torch.argsort(torch.rand(32768, n_samples_x * n_samples_x, device=device, generator=g), dim=-1).int()
        _optix_env_shade_func._random_perm[key] = t
    return t


class _optix_env_shade_func(torch.autograd.Function):
    _random_perm = {}          # (device type, device index, n_samples_x) -> int32 [32768, S]
    # module defaults of the per-context switches (OptiXContext.cache_visibility / .pixel_index_offset)
    cache_visibility = True
    pixel_index_offset = 0

    @staticmethod
    def _switches(optix_ctx):
        cv = getattr(optix_ctx, 'cache_visibility', None)
        off = getattr(optix_ctx, 'pixel_index_offset', None)
        return (_optix_env_shade_func.cache_visibility if cv is None else bool(cv),
                _optix_env_shade_func.pixel_index_offset if off is None else int(off))

    @staticmethod
    def forward(ctx, optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, BSDF,
                n_samples_x, rnd_seed, shadow_scale):
        ctx.set_materialize_grads(False)
        _rnd_seed = np.random.randint(2**31) if rnd_seed is None else rnd_seed
        perms = _perms_for(n_samples_x, ro.device)
        w = optix_ctx.cpp_wrapper
        cache_vis, off = _optix_env_shade_func._switches(optix_ctx)
        # the device-resident seed counter is SNAPSHOT here: the caller advances it right after this call (render.py:116),
        # long before backward runs, and backward must repeat the forward's samples (same slots, same cached visibility bits)
        so = getattr(optix_ctx, 'seed_offset', None)
        adv = int(getattr(optix_ctx, 'seed_advance', 0) or 0)
        if so is not None and rnd_seed is not None and adv:
            # the launch itself snapshots the counter and advances it (render.py:116 `rnd_seed += 1`): no clone / add kernels
            seed_snap = torch.empty_like(so)
            a = _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
                           BSDF, n_samples_x, _rnd_seed, shadow_scale, off, so)
            a.rnd_seed_snapshot, a.rnd_seed_advance = seed_snap.data_ptr(), adv
        else:
            seed_snap = so.clone() if (so is not None and rnd_seed is not None) else so
            a = _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
                           BSDF, n_samples_x, _rnd_seed, shadow_scale, off, seed_snap)
        N, H, W = ro.shape[0], ro.shape[1], ro.shape[2]
        # independent storages like the reference's two torch::zeros (torch_bindings.cpp:148-149): views of one packed
        # buffer would make any in-place op on an output an autograd error; the library zero-fills them
        diff = torch.empty(N, H, W, 3, dtype=torch.float32, device=ro.device)
        spec = torch.empty(N, H, W, 3, dtype=torch.float32, device=ro.device)
        a.diff, a.spec = diff.data_ptr(), spec.data_ptr()
        vis = None
        if rnd_seed is not None and cache_vis:
            words = (n_samples_x * n_samples_x + 31) // 32
            vis = torch.empty(N * H * W * 2 * words, dtype=torch.int32, device=ro.device)
            a.vis_cache = vis.data_ptr()
        hook = getattr(optix_ctx, 'split_hook', None)
        if hook is not None:
            a.phase = 1
            _lib.check(w.lib.nvdr_env_shade_fwd(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'env_shade_fwd')
            hook()                  # (may change the current stream: a capture ends here and the next one begins)
            a.phase = 2
        _lib.check(w.lib.nvdr_env_shade_fwd(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'env_shade_fwd')
        sid = ctypes.c_uint64()
        _lib.check(w.lib.nvdr_env_shade_stream_id(w.handle, ctypes.byref(sid)), 'env_shade_stream_id')
        ctx.stream_id = int(sid.value) if rnd_seed is not None else 0   # a decorrelated backward draws new samples
        ctx.save_for_backward(mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols)
        ctx.optix_ctx = optix_ctx
        ctx.BSDF = BSDF
        ctx.n_samples_x = n_samples_x
        ctx.rnd_seed = rnd_seed
        ctx.shadow_scale = shadow_scale
        ctx.vis = vis
        ctx.pixel_index_offset = off
        ctx.seed_snap = seed_snap
        ctx.bvh_geom = w._geom
        return diff, spec

    @staticmethod
    def backward(ctx, diff_grad, spec_grad):
        optix_ctx = ctx.optix_ctx
        _rnd_seed = np.random.randint(2**31) if ctx.rnd_seed is None else ctx.rnd_seed
        mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols = ctx.saved_tensors
        perms = _perms_for(ctx.n_samples_x, ro.device)
        w = optix_ctx.cpp_wrapper
        a = _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
                       ctx.BSDF, ctx.n_samples_x, _rnd_seed, ctx.shadow_scale, ctx.pixel_index_offset, ctx.seed_snap)
        N, H, W = ro.shape[0], ro.shape[1], ro.shape[2]
        dev = ro.device
        if diff_grad is None:
            diff_grad = torch.zeros(N, H, W, 3, dtype=torch.float32, device=dev)
        if spec_grad is None:
            spec_grad = torch.zeros(N, H, W, 3, dtype=torch.float32, device=dev)
        diff_grad, spec_grad = diff_grad.contiguous(), spec_grad.contiguous()
        a.diff_grad = _lib.tensor_view(diff_grad, lead=False)
        a.spec_grad = _lib.tensor_view(spec_grad, lead=False)
        # independent storages (a leaf's .grad may keep one of them alive; round 1 handed out views of one 4x buffer)
        gb_pos_grad, gb_normal_grad, gb_kd_grad, gb_ks_grad = (torch.empty(N, H, W, 3, dtype=torch.float32, device=dev) for _ in range(4))
        light_grad = torch.empty(light.shape[0], light.shape[1], 3, dtype=torch.float32, device=dev)
        a.gb_pos_grad, a.gb_normal_grad = gb_pos_grad.data_ptr(), gb_normal_grad.data_ptr()
        a.gb_kd_grad, a.gb_ks_grad, a.light_grad = gb_kd_grad.data_ptr(), gb_ks_grad.data_ptr(), light_grad.data_ptr()
        # the cached bits / the stored ray stream are only valid while the context still holds the geometry of the
        # forward pass (the library additionally checks that no other launch has overwritten the stream)
        if ctx.bvh_geom is w._geom:
            a.reuse_stream_id = ctx.stream_id
            if ctx.vis is not None:
                a.vis_cache = ctx.vis.data_ptr()
        _lib.check(w.lib.nvdr_env_shade_bwd(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'env_shade_bwd')
        if light.shape[-1] == 1:
            light_grad = light_grad.sum(-1, keepdim=True)
        return (None, None, None, gb_pos_grad, gb_normal_grad, None, gb_kd_grad, gb_ks_grad, light_grad,
                None, None, None, None, None, None, None)


def set_permutation_table(n_samples_x, perms):
    """Install a specific permutation table (int32 [NP, n_samples_x^2], on the GPU) -- parity tests only."""
    d = perms.device
    _optix_env_shade_func._random_perm[(d.type, d.index if d.index is not None else torch.cuda.current_device(), n_samples_x)] = perms


def set_pixel_index_offset(offset, optix_ctx=None):
    """Data-parallel shards: offset added to the linear pixel index that seeds the RNG (first_view * H * W); per context
    when one is given, otherwise the module default."""
    if optix_ctx is not None:
        optix_ctx.pixel_index_offset = int(offset)
    else:
        _optix_env_shade_func.pixel_index_offset = int(offset)


def optix_env_shade(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols,
                    BSDF='pbr', n_samples_x=8, rnd_seed=None, shadow_scale=1.0):
    iBSDF = ['pbr', 'diffuse', 'white'].index(BSDF)  # ordering as in the reference (ops.py:136)
    return _optix_env_shade_func.apply(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf,
                                       rows, cols, iBSDF, n_samples_x, rnd_seed, shadow_scale)


# ----------------------------------------------------------------------------------------------

class _bilateral_denoiser_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, col, nrm, zdz, sigma):
        for name, t in (('col', col), ('nrm', nrm), ('zdz', zdz)):
            _lib.require_cuda_f32(t, name)
        ctx.save_for_backward(col, nrm, zdz)
        ctx.sigma = sigma
        lib = _lib.load()
        N, H, W = col.shape[0], col.shape[1], col.shape[2]
        out = torch.empty(N, H, W, 4, dtype=torch.float32, device=col.device)
        vc, vn, vz = _lib.tensor_view(col), _lib.tensor_view(nrm), _lib.tensor_view(zdz)
        _lib.check(lib.nvdr_bilateral_denoiser_fwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), float(sigma),
                                                   _lib.ptr(out), _lib.stream_ptr()), 'bilateral_denoiser_fwd')
        return out

    @staticmethod
    def backward(ctx, out_grad):
        col, nrm, zdz = ctx.saved_tensors
        lib = _lib.load()
        N, H, W = col.shape[0], col.shape[1], col.shape[2]
        col_grad = torch.empty(N, H, W, 3, dtype=torch.float32, device=col.device)
        vc, vn, vz, vg = _lib.tensor_view(col), _lib.tensor_view(nrm), _lib.tensor_view(zdz), _lib.tensor_view(out_grad)
        _lib.check(lib.nvdr_bilateral_denoiser_bwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), float(ctx.sigma),
                                                   ctypes.byref(vg), _lib.ptr(col_grad), _lib.stream_ptr()),
                   'bilateral_denoiser_bwd')
        return col_grad, None, None, None


class _bilateral_denoiser_pair_func(torch.autograd.Function):
    """Two images with the same guides in one pass (the diffuse and the specular light of shade(), render.py:120-121): the weights
    are evaluated once per tap.  Outputs bit-identical to two _bilateral_denoiser_func calls."""

    @staticmethod
    def forward(ctx, col_a, col_b, nrm, zdz, sigma):
        for name, t in (('col_a', col_a), ('col_b', col_b), ('nrm', nrm), ('zdz', zdz)):
            _lib.require_cuda_f32(t, name)
        ctx.save_for_backward(col_a, nrm, zdz)
        ctx.sigma = sigma
        N, H, W = col_a.shape[0], col_a.shape[1], col_a.shape[2]
        out = torch.empty(2, N, H, W, 4, dtype=torch.float32, device=col_a.device)
        va, vb, vn, vz = _lib.tensor_view(col_a), _lib.tensor_view(col_b), _lib.tensor_view(nrm), _lib.tensor_view(zdz)
        _lib.check(_lib.load().nvdr_bilateral_denoiser_pair_fwd(ctypes.byref(va), ctypes.byref(vb), ctypes.byref(vn), ctypes.byref(vz), float(sigma),
                                                                _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.stream_ptr()), 'bilateral_denoiser_pair_fwd')
        return out[0], out[1]

    @staticmethod
    def backward(ctx, grad_a, grad_b):
        col, nrm, zdz = ctx.saved_tensors
        N, H, W = col.shape[0], col.shape[1], col.shape[2]
        g = torch.empty(2, N, H, W, 3, dtype=torch.float32, device=col.device)
        vc, vn, vz = _lib.tensor_view(col), _lib.tensor_view(nrm), _lib.tensor_view(zdz)
        ga, gb = _lib.tensor_view(grad_a), _lib.tensor_view(grad_b)
        _lib.check(_lib.load().nvdr_bilateral_denoiser_pair_bwd(ctypes.byref(vc), ctypes.byref(vn), ctypes.byref(vz), float(ctx.sigma), ctypes.byref(ga),
                                                                ctypes.byref(gb), _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.stream_ptr()),
                   'bilateral_denoiser_pair_bwd')
        return g[0], g[1], None, None, None


def bilateral_denoiser(col, nrm, zdz, sigma):
    col_w = _bilateral_denoiser_func.apply(col, nrm, zdz, sigma)
    return col_w[..., 0:3] / col_w[..., 3:4]


# ----------------------------------------------------------------------------------------------
# additive ray-query helpers (no reference counterpart): test hooks and the G-buffer producer

def trace_visibility(optix_ctx, ro, rd, count=False):
    """uint8 [R]: 1 where the ray (ro, rd) hits nothing for t in (0, 1e16).  With count=True also returns
    (box tests, triangle tests) summed over all rays."""
    w = optix_ctx.cpp_wrapper
    _lib.require_cuda_f32(ro, 'ro')
    _lib.require_cuda_f32(rd, 'rd')
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    R = ro.shape[0]
    vis = torch.empty(R, dtype=torch.uint8, device=ro.device)
    cnt = torch.zeros(2, dtype=torch.int64, device=ro.device) if count else None
    _lib.check(w.lib.nvdr_trace_visibility(w.handle, _lib.ptr(ro), _lib.ptr(rd), R, _lib.ptr(vis), _lib.ptr(cnt),
                                           _lib.stream_ptr()), 'trace_visibility')
    return (vis, cnt) if count else vis


def trace_visibility_wide(optix_ctx, ro, rd, count=False):
    """trace_visibility through the PRODUCTION shadow-ray kernel (wide nodes, persistent wavefronts); test hook.
    With count=True the counting build of the same kernel runs and (box tests, triangle tests, rays, node steps) are returned too."""
    w = optix_ctx.cpp_wrapper
    _lib.require_cuda_f32(ro, 'ro')
    _lib.require_cuda_f32(rd, 'rd')
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    vis = torch.empty(ro.shape[0], dtype=torch.uint8, device=ro.device)
    if not count:
        _lib.check(w.lib.nvdr_trace_visibility_wide(w.handle, _lib.ptr(ro), _lib.ptr(rd), ro.shape[0], _lib.ptr(vis), _lib.stream_ptr()),
                   'trace_visibility_wide')
        return vis
    cnt = torch.zeros(_lib.COUNTERS_LEN, dtype=torch.int64, device=ro.device)
    _lib.check(w.lib.nvdr_trace_visibility_wide_counted(w.handle, _lib.ptr(ro), _lib.ptr(rd), ro.shape[0], _lib.ptr(vis), _lib.ptr(cnt),
                                                        _lib.stream_ptr()), 'trace_visibility_wide_counted')
    c = cnt.cpu()
    trace_visibility_wide.last_counters = c        # (the whole block, nvdr_hip.h NVDR_COUNTERS_*: tests and tools read more of it)
    return vis, (int(c[0]), int(c[1]), int(c[2]), int(c[_lib.COUNTERS_BVH2 + 3]))


def trace_closest(optix_ctx, ro, rd):
    """Closest hit: (t [R] (<0 miss), triangle index [R] int32 (-1 miss), barycentrics [R,2])."""
    w = optix_ctx.cpp_wrapper
    _lib.require_cuda_f32(ro, 'ro')
    _lib.require_cuda_f32(rd, 'rd')
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    R = ro.shape[0]
    t = torch.empty(R, dtype=torch.float32, device=ro.device)
    tri = torch.empty(R, dtype=torch.int32, device=ro.device)
    uv = torch.empty(R, 2, dtype=torch.float32, device=ro.device)
    _lib.check(w.lib.nvdr_trace_closest(w.handle, _lib.ptr(ro), _lib.ptr(rd), R, _lib.ptr(t), _lib.ptr(tri), _lib.ptr(uv),
                                        _lib.stream_ptr()), 'trace_closest')
    return t, tri, uv


_GB_CHANNELS = (('rast', 4), ('rast_db', 4), ('gb_pos', 3), ('gb_geometric_normal', 3), ('gb_normal', 3), ('gb_tangent', 3),
                ('gb_texc', 2), ('gb_texc_deriv', 4), ('gb_depth', 2))


def render_gbuffer(optix_ctx, mesh, mvp, cam, resolution):
    """G-buffers of N views from primary rays through the context's BVH: what render_layer takes from nvdiffrast's
    rasterize + interpolate (render/render.py:208-234).  mesh: dict with v_pos, t_pos_idx, v_nrm, t_nrm_idx, v_tng, t_tng_idx,
    v_tex, t_tex_idx (GPU tensors; the BVH must have been built from v_pos / t_pos_idx); mvp [N,4,4]; cam [N,4,3] =
    (eye, U, V, W) per view (scene.camera_rays); resolution (H, W).  Returns a dict of contiguous NHWC tensors:
    rast (u, v, z/w, id + 1), rast_db, gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_texc, gb_texc_deriv, gb_depth."""
    w = optix_ctx.cpp_wrapper
    H, W = int(resolution[0]), int(resolution[1])
    a = _lib.NvdrGbufferArgs()
    keep = []
    for k, dt in (('v_pos', torch.float32), ('t_pos_idx', torch.int32), ('v_nrm', torch.float32), ('t_nrm_idx', torch.int32),
                  ('v_tng', torch.float32), ('t_tng_idx', torch.int32), ('v_tex', torch.float32), ('t_tex_idx', torch.int32)):
        t = mesh[k]
        _lib.require_cuda_f32(t, k, dt)
        t = t.contiguous()
        keep.append(t)
        setattr(a, k, t.data_ptr())
    a.n_tris = mesh['t_pos_idx'].shape[0]
    for k in ('t_nrm_idx', 't_tng_idx', 't_tex_idx'):
        if mesh[k].shape[0] != a.n_tris:
            raise RuntimeError('%s must have one row per triangle' % k)
    _lib.require_cuda_f32(mvp, 'mvp')
    _lib.require_cuda_f32(cam, 'cam')
    mvp, cam = mvp.contiguous(), cam.contiguous()
    N = mvp.shape[0]
    if tuple(mvp.shape) != (N, 4, 4) or tuple(cam.shape) != (N, 4, 3):
        raise RuntimeError('mvp must be [N,4,4] and cam [N,4,3] (got %s, %s)' % (tuple(mvp.shape), tuple(cam.shape)))
    a.mvp, a.cam, a.n, a.h, a.w = mvp.data_ptr(), cam.data_ptr(), N, H, W
    out = {}
    for k, ch in _GB_CHANNELS:
        out[k] = torch.empty(N, H, W, ch, dtype=torch.float32, device=mvp.device)
        setattr(a, k, out[k].data_ptr())
    _lib.check(w.lib.nvdr_render_gbuffer(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'render_gbuffer')
    return out


def env_shade_forward_with_bits(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols,
                                BSDF='pbr', n_samples_x=8, rnd_seed=0, shadow_scale=1.0):
    """Test hook: one forward launch that also returns the visibility bits the kernels used -- int32 [N*H*W, 2, ceil(S/32)],
    bit s of plane 0 / 1 = the light- / BSDF-sampled shadow ray of STRATUM s is occluded (nvdr_env_shade_args.vis_cache)."""
    perms = _perms_for(n_samples_x, ro.device)
    w = optix_ctx.cpp_wrapper
    a = _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
                   ['pbr', 'diffuse', 'white'].index(BSDF), n_samples_x, rnd_seed, shadow_scale,
                   _optix_env_shade_func._switches(optix_ctx)[1], getattr(optix_ctx, 'seed_offset', None))
    N, H, W = ro.shape[0], ro.shape[1], ro.shape[2]
    diff = torch.empty(N, H, W, 3, dtype=torch.float32, device=ro.device)
    spec = torch.empty_like(diff)
    words = (n_samples_x * n_samples_x + 31) // 32
    bits = torch.zeros(N * H * W, 2, words, dtype=torch.int32, device=ro.device)
    a.diff, a.spec, a.vis_cache = diff.data_ptr(), spec.data_ptr(), bits.data_ptr()
    _lib.check(w.lib.nvdr_env_shade_fwd(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'env_shade_fwd(bits)')
    return diff, spec, bits


def env_shade_traversal_counts(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols,
                               BSDF='pbr', n_samples_x=8, rnd_seed=0, shadow_scale=1.0):
    """Run the COUNTING build of the forward kernel once: returns (covered pixels, box tests, triangle tests,
    rays traversed).  Rays traversed is below 2*S*pixels: samples under the shading horizon are never traced.
    env_shade_traversal_counts.bvh2 = (node visits, triangle tests, rays) of the canonical binary walk over the same rays."""
    perms = _perms_for(n_samples_x, ro.device)
    w = optix_ctx.cpp_wrapper
    a = _fill_args(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms,
                   ['pbr', 'diffuse', 'white'].index(BSDF), n_samples_x, rnd_seed, shadow_scale,
                   _optix_env_shade_func._switches(optix_ctx)[1], getattr(optix_ctx, 'seed_offset', None))
    N, H, W = ro.shape[0], ro.shape[1], ro.shape[2]
    diff = torch.empty(N, H, W, 3, dtype=torch.float32, device=ro.device)
    spec = torch.empty_like(diff)
    cnt = torch.zeros(_lib.COUNTERS_LEN, dtype=torch.int64, device=ro.device)
    a.diff, a.spec, a.counters = diff.data_ptr(), spec.data_ptr(), cnt.data_ptr()
    _lib.check(w.lib.nvdr_env_shade_fwd(w.handle, ctypes.byref(a), _lib.stream_ptr()), 'env_shade_fwd(count)')
    npx = ctypes.c_int64()
    _lib.check(w.lib.nvdr_env_shade_last_pixel_count(w.handle, ctypes.byref(npx), _lib.stream_ptr()), 'pixel_count')
    c = cnt.cpu()
    env_shade_traversal_counts.balance = (int(c[3]), int(c[4]), int(c[5]))   # sum, max of per-wave ticks (100 MHz), waves
    wt = c[8:8 + 2 * min(int(c[5]), 8192)].view(-1, 2).clone()
    env_shade_traversal_counts.wave_xcd = (wt[:, 1] >> 56) & 7                                  # XCD every wavefront ran on
    wt[:, 1] &= (1 << 56) - 1
    env_shade_traversal_counts.wave_ticks = wt                                                   # (begin, end) per wavefront, 100 MHz ticks
    env_shade_traversal_counts.clock_mhz = 100.0 * int(c[6]) / max(int(c[3]), 1)     # shader clock the waves ran at
    env_shade_traversal_counts.xcd_mask = int(c[7])
    env_shade_traversal_counts.bvh2 = tuple(int(v) for v in c[_lib.COUNTERS_BVH2:_lib.COUNTERS_BVH2 + 3])
    env_shade_traversal_counts.node_steps = int(c[_lib.COUNTERS_BVH2 + 3])      # node visits of the production walk
    # cycles by loop phase from the two phase-clock builds (nvdr_hip.h NVDR_COUNTERS_PHASES): [refill, node step, queue rounds, batches, iterations,
    # node-step iterations, total cycles, wavefronts] and [refill, fetch, box, stack, queue rounds, batches, total cycles, iterations]
    env_shade_traversal_counts.phases = [int(v) for v in c[_lib.COUNTERS_PHASES:_lib.COUNTERS_PHASES + 8]]
    env_shade_traversal_counts.phases_split = [int(v) for v in c[_lib.COUNTERS_PHASES + 8:_lib.COUNTERS_PHASES + 16]]
    env_shade_traversal_counts.leaf_batches = (int(c[_lib.COUNTERS_BVH2 + 4]), int(c[_lib.COUNTERS_BVH2 + 5]))   # triangle-test batches, lanes they filled
    return int(npx.value), int(c[0]), int(c[1]), int(c[2])
