"""Minimal optimisation step around the direct-lighting hot path (SURVEY 8 f1 + f2).

One iteration mirrors what optimize_mesh does per step in the reference (train.py:385-476), reduced to
the parts that touch the hot path:

    lgt.update_pdf()                                     train.py:422      -> csrc/light.hip
    optix_build_bvh(rebuild=1)                           dlmesh.py:50      -> csrc/bvh.hip
    prepare_shading_normal                               render.py:99      -> csrc/renderutils.hip
    optix_env_shade (fwd)                                render.py:113     -> csrc/env_shade.hip
    bilateral filter x2 on (light, normal, depth)        render.py:120-121 -> csrc/denoise.hip
    shaded = diffuse * kd * (1 - metal) + specular       render.py:126-127 -> csrc/renderutils.hip (shade_composite)
    image_loss('l1', 'log_srgb')                         train.py:51-66    -> csrc/renderutils.hip
    backward through all of it (env-shade re-traces)     train.py:438
    gradient all-reduce (new: one view per GPU)          --                -> parallel.py
    Adam on kd texture, ks, light                        train.py:452-461

The trained set is the reference's (train.py:171-192, geometry/dlmesh.py:28-38): kd, ks and normal TEXTURES (texture_res of
the config, 1024^2 each), the light probe and -- optimize_geometry=True, lock_pos false -- the vertex positions:

    texture lookups at gb_texc                           render.py:61-68   -> csrc/mesh.hip (nearest texel, one launch each way)
    perturbed normal -> prepare_shading_normal           render.py:85-99   -> csrc/renderutils.hip (shading_frame, backward included)
    auto_normals, compute_tangents                       dlmesh.py:52-54   -> csrc/mesh.hip (optimize_geometry)
    rasterize + interpolate                              render.py:208-234 -> csrc/gbuffer.hip, adjoint csrc/mesh.hip (optimize_geometry)

What replaces the parts that do not exist on ROCm: the G-buffer comes from ONE HIP kernel (primary rays through the same BVH +
the attribute interpolation / tangents / (z/w, |dz|) of render_layer) instead of nvdiffrast; its adjoint carries the gradients of
gb_pos / gb_normal / gb_tangent / the face normal back to the vertices (attribute values and barycentrics), but coverage is not
differentiated (dr.antialias, render.py:290, has no counterpart).  Textures are sampled at the nearest texel instead of
dr.texture's trilinear mip filter.  With locked geometry (configs/bob.json:14) the G-buffer of the fixed benchmark views is
rendered once; with optimize_geometry it is re-rendered every iteration from the moving vertices, like the reference does.
material_set='r3' keeps round 3's reduced set (kd texture at the asset's resolution + ONE global ks vector, no normal map) for A/B.
"""
import math
import os

import torch

from . import optixutils as ou
from . import renderutils as ru
from . import scene as sc
from .denoiser import BilateralDenoiser, _safe_normalize
from .light import EnvironmentLight
from .optim import FusedAdam
from . import mesh as mesh_ops
from . import render as rd
from . import _lib



def _capture_graph(g, **kw):
    """torch.cuda.graph(g, ...) with THREAD-LOCAL capture errors.  The default ('global') makes every "unsafe" HIP call of ANY thread an error while a
    stream captures -- and the process group's watchdog thread polls its pending collectives with hipEventQuery: a capture that begins while a
    collective of the warm-up iterations is still on the watchdog's list died, once in ~25 runs of the several-rank schedule, with "operation not
    permitted when stream is capturing" (round 6, sessions 29-31; `tools/capture_mode_probe.py` shows it deterministically).  Thread-local mode keeps
    the check for the capturing thread, which is the one whose calls end up in the graph."""
    return torch.cuda.graph(g, capture_error_mode='thread_local', **kw)


class _gather_rows(torch.autograd.Function):
    """tex[idx] with an index_add_ (atomic) backward.  torch's generic advanced-indexing backward sorts the
    indices and serialises on duplicates: 57 ms per iteration here, 20x the whole hot path."""

    @staticmethod
    def forward(ctx, tex, idx):
        ctx.save_for_backward(idx)
        ctx.n = tex.shape[0]
        return tex.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        out = torch.zeros(ctx.n, g.shape[1], dtype=g.dtype, device=g.device)
        out.index_add_(0, idx, g.contiguous())
        return out, None


class _lookup_rows(torch.autograd.Function):
    """out[i] = idx[i] >= 0 ? tex[idx[i]] : 0 in one launch (csrc/renderutils.hip gather_rows), backward = memset + one atomic pass:
    three launches instead of the six of zeros / index_select / index_copy and their adjoints."""

    @staticmethod
    def forward(ctx, tex, idx):
        from . import _lib
        ctx.save_for_backward(idx)
        ctx.rows = tex.shape[0]
        tex = tex.contiguous()
        out = torch.empty(idx.numel(), tex.shape[1], dtype=torch.float32, device=tex.device)
        _lib.check(_lib.load().nvdr_gather_rows_fwd(_lib.ptr(tex), _lib.ptr(idx), idx.numel(), tex.shape[1], _lib.ptr(out), _lib.stream_ptr()), 'gather_rows_fwd')
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        idx, = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty(ctx.rows, g.shape[1], dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().nvdr_gather_rows_bwd(_lib.ptr(g), _lib.ptr(idx), idx.numel(), g.shape[1], ctx.rows, _lib.ptr(out), _lib.stream_ptr()), 'gather_rows_bwd')
        return out, None


class _broadcast_pixels(torch.autograd.Function):
    """x[C] -> [N,H,W,C] stride-0 view, with a column sum as backward that does not go through torch's generic
    reduction (summing [262144, 3] over its long dimension took 91 us: one wavefront per output column)."""

    @staticmethod
    def forward(ctx, x, N, H, W):
        return x.view(1, 1, 1, -1).expand(N, H, W, x.numel())

    @staticmethod
    def backward(ctx, g):
        C = g.shape[-1]
        P = g.numel() // C
        k = 256 if P % 256 == 0 else 1
        return g.reshape(P // k, k * C).sum(0).view(k, C).sum(0), None, None, None


class DirectLightingStep:
    def __init__(self, mesh_name='bob', res=512, n_samples_x=8, view=0, n_views=8, device='cuda', env='E1',
                 probe_res=256, denoise=True, retrace_backward=False, pixel_index_offset=0, subdiv=0, lr=0.01, fused=True,
                 denoiser_demodulate=True, light_grad_scale=64.0, use_graph=False, material_set='full', tex_res=1024,
                 optimize_geometry=False, lr_pos=None, lr_light=None, perturb_pos=0.0, ks_min=(0.0, 0.08, 0.0), ks_max=(0.0, 1.0, 1.0),
                 perturbed_nrm=True, exchange_mode='dense', pipeline=True, force_exchange=False, union_views=None, build_mode=None, rebuild_every=1):
        self.dev = torch.device(device)
        self.res, self.n, self.view = res, n_samples_x, view     # view: an index or a list of indices (a batch of views)
        self.pixel_index_offset = pixel_index_offset
        # retrace_backward=False (default): the forward pass and its backward pass draw the same samples (train.py:547 `decorrelated = False`,
        # render.py:112-116: one rnd_seed for both), so the backward pass replays the forward's visibility bits instead of traversing the same
        # rays again -- per-pixel gradients bit-identical (tests/test_gpu_env_shade.py), one traversal launch less.  True = what the
        # reference's backward does (optixTrace again): the iteration bench.py's `value` is defined on.
        self.retrace_backward = retrace_backward
        # perturbed_nrm=False: FLAGS.no_perturbed_nrm (configs/spot_metal.json:20, render.py:92-93): no normal-map lookup; the normal texture
        # stays in the optimizer's list (train.py:185-197) but never receives a gradient, i.e. is never updated -- it is left out of the set here
        self.perturbed_nrm = bool(perturbed_nrm)
        # Several ranks (parallel.GradientExchange).  exchange_mode: 'dense' (default since round 6: what 'auto' settles on for the benchmark views, without
        # its probe rounds) = the whole texture bucket is all-reduced; 'sparse' = only the
        # 768-byte tiles some rank's pixels touched (the dense bucket when more than half are); 'auto' = sparse when a periodic probe finds at
        # most a quarter of the tiles touched, plainly dense otherwise (the benchmark views touch half of them: auto runs dense there).  pipeline: the
        # texture chunk's reduce runs under the NEXT iteration's geometry stage and is waited for in front of the texture lookup.
        # force_exchange: run the several-rank schedule with ONE rank (the fixed cost of the path; a one-rank RCCL pass when a process group exists).
        if exchange_mode not in ('auto', 'dense', 'sparse'):
            raise ValueError("exchange_mode must be 'auto', 'dense' or 'sparse'")
        self.exchange_mode, self.pipeline, self.force_exchange = exchange_mode, bool(pipeline), bool(force_exchange)
        # rebuild_every (trained geometry only): K > 1 = the tree is REBUILT every K-th iteration and REFITTED (rebuild=0: the topology of the
        # last rebuild, new boxes, the eight-wide collapse redone) in between.  Visibility and closest hits are exact with any valid tree; the
        # vertices move by a learning-rate step per iteration, so a refitted tree degrades slowly and a rebuild every few iterations
        # restores it -- the practice OptiX documents for dynamic geometry.  1 = what the reference does (geometry/dlmesh.py:50 passes
        # rebuild=1 every iteration) and the default.  Locked geometry is rebuilt every iteration whatever K (as the reference does).
        self.rebuild_every = max(1, int(rebuild_every))
        self._iter = 0
        # union_views (with force_exchange): the views of the WHOLE batch; the tile flags of this one rank are OR-ed with the tiles those views
        # touch, so that the one-rank run compacts, sends and scatters the bytes the several-rank run would (bench.py one_view)
        self._union_views = list(union_views) if (union_views is not None and force_exchange) else None
        self.measure_exposed, self._exposed_events, self._stage1_events = False, [], []
        self._stage1_ready, self._gb_live, self._pending = False, None, False
        self.split_stage2 = False         # set by _capture: locked geometry under the pipelined several-rank schedule (stage 2 in two graphs)
        self.fused = fused
        self.pair_filter = _lib.tuning_env('NVDR_PAIR_FILTER', '1') != '0'      # (A/B switch of the harness)
        self.fused_loss = _lib.tuning_env('NVDR_FUSED_LOSS', '1') != '0'        # (A/B switch: composite + loss + adjoints in one launch, renderutils.shade_composite_loss)
        self.denoiser_demodulate = denoiser_demodulate    # FLAGS.denoiser_demodulate (train.py:525, default True)
        self.light_grad_scale = light_grad_scale          # lgt.base.grad *= 64 (train.py:439-440)
        self.total_views = n_views if isinstance(n_views, int) else len(n_views)
        if material_set not in ('full', 'r3'):
            raise ValueError("material_set must be 'full' or 'r3'")
        if optimize_geometry and material_set != 'full':
            raise ValueError('optimize_geometry needs material_set="full"')
        self.material_set, self.optimize_geometry, self.tex_res_train = material_set, bool(optimize_geometry), int(tex_res)
        # use_graph: capture the iteration in HIP graphs (one submit instead of ~110 launches) once it has run a few times
        # eagerly.  What makes that legal: nothing on the path synchronises the host or allocates after warm-up, and the
        # random seed of shade() lives in device memory (OptiXContext.seed_offset), so a replay draws fresh samples.
        self.use_graph = use_graph
        self._graphs = None
        self._eager_steps = 0
        self.force_eager = False          # set once at the END of a run to leave graph mode for good (p.grad then belongs to eager)
        mesh = sc.load_mesh(mesh_name, device='cpu')
        if subdiv:
            mesh = sc.subdivide_mesh(mesh, subdiv, cache_key=mesh_name)       # a DMTet-sized stand-in (positions and uvs subdivided, new normals / tangents)
        self.mesh = {k: (v.to(self.dev) if isinstance(v, torch.Tensor) else v) for k, v in mesh.items()}
        self.ctx = ou.OptiXContext()
        ou.optix_build_bvh(self.ctx, self.mesh['v_pos'], self.mesh['t_pos_idx'], rebuild=1)
        self._build_mode = build_mode

        # ---- G-buffers of this rank's views in ONE kernel launch (csrc/gbuffer.hip): primary rays through the same BVH +
        # the attribute interpolation, face normal, tangents and (z/w, |dz|) pair of render_layer (render.py:208-234); stands
        # in for nvdiffrast's rasterize + interpolate.  The views are stacked along N like the reference's batch (configs/bob.json:8).
        H = W = res
        views = list(view) if isinstance(view, (list, tuple)) else [view]
        self.views = views
        cams = [sc.camera(vw, n_views) for vw in views]
        self.mvp = torch.stack([c[1] for c in cams]).to(self.dev)
        self.cam = torch.stack([sc.camera_rays(c[0]) for c in cams]).to(self.dev)
        self.nv = len(views)
        self.view_pos = torch.stack([c[2] for c in cams]).to(self.dev)[:, None, None, :].contiguous()    # [V,1,1,3]
        self.topo = mesh_ops.MeshTopology(self.mesh['t_pos_idx'], self.mesh['v_pos'].shape[0], self.mesh['v_tex'], self.mesh['t_tex_idx'])
        self._set_gbuffer(ou.render_gbuffer(self.ctx, self.mesh, self.mvp, self.cam, (H, W)))
        if material_set == 'r3':
            # texel addresses of the kd lookup (fixed: geometry is locked, configs/bob.json:14); nearest texel of a trainable
            # texture instead of dr.texture's trilinear mip lookup (outside the path)
            R = self.mesh['kd_tex'].shape[0]
            tc = self.gb_texc
            ix = (tc[..., 0] * R).long().clamp(0, R - 1)
            iy = ((1.0 - tc[..., 1]) * R).long().clamp(0, R - 1)
            self.texel = (iy * R + ix).view(-1)
            self.tex_res = R
            # only covered pixels look the texture up (the background would pile ~200k duplicates on one texel)
            self.cov = self.mask.view(-1).nonzero().view(-1)
            self.texel_cov = self.texel[self.cov].contiguous()
            self.texel_or_none = torch.where(self.mask.view(-1) > 0, self.texel, torch.full_like(self.texel, -1)).to(torch.int32).contiguous()
        kd_true = self.mesh['kd_tex'].reshape(-1, 3)

        # ---- reference ("true") parameters -> target image; trainable parameters start elsewhere
        self.denoiser = BilateralDenoiser(influence=1.0) if denoise else None
        light_true = EnvironmentLight(sc.env_map(env, probe_res).to(self.dev))
        ks_true = self.mesh['ks'].clone()
        Rt = self.mesh['kd_tex'].shape[0]
        # the global seed counter of render.py:19,112-116 -- kept in DEVICE memory and added to rnd_seed by the kernels
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.ctx.seed_offset = self.seed_dev
        self.ctx.seed_advance = 1        # render.py:116 `rnd_seed += 1`, done by the env-shade launch itself
        with torch.no_grad():
            if material_set == 'r3':
                self.target = self._render(kd_true, ks_true, light_true).detach()
            else:       # the asset's own material: kd texture, constant ks (data/bob/bob_tri.mtl:3), no normal map
                self.target = self._render_full(self.mesh['kd_tex'].contiguous(), ks_true.view(1, 1, 3).expand(Rt, Rt, 3).contiguous(),
                                                torch.tensor([0.0, 0.0, 1.0], device=self.dev).repeat(Rt, Rt, 1).contiguous() if self.perturbed_nrm else None,
                                                light_true).detach()
        self._ks_min = torch.tensor(list(ks_min), device=self.dev)
        self._ks_max = torch.tensor(list(ks_max), device=self.dev)
        self.light = EnvironmentLight(torch.full((probe_res, probe_res, 3), 0.5, device=self.dev).requires_grad_(True))
        lr_light = 3.0 * lr if lr_light is None else lr_light            # learning_rate_lgt = learning_rate * 3.0 (train.py:338)
        lr_pos = lr if lr_pos is None else lr_pos                          # train.py:336
        if material_set == 'r3':
            self.kd_tex = torch.nn.Parameter(torch.full_like(kd_true, 0.5))
            self.ks = torch.nn.Parameter(torch.tensor([0.0, 0.5, 0.0], device=self.dev))
            self.params = [self.kd_tex, self.ks, self.light.base]
            names = ['kd', 'ks', 'light']
            clamps = [(0.0, 1.0), (None, 1.0, self._ks_min), (0.0, None)]
            lr_scales, grad_scales, norm3 = [1.0, 1.0, 1.0], [1.0, 1.0, light_grad_scale], [False, False, False]
        else:
            # initial_guess_material with random_textures (train.py:171-192): kd flat, ks uniform random per texel inside
            # [ks_min, ks_max] (ksR in [0, 0.01]), the normal map (0, 0, 1); Texture2D.clamp_ bounds per channel
            R = self.tex_res_train
            gen = torch.Generator().manual_seed(1234)
            ks0 = torch.rand(R, R, 3, generator=gen)
            lo = torch.tensor([0.0, ks_min[1], ks_min[2]])
            hi = torch.tensor([0.01, ks_max[1], ks_max[2]])
            self.kd_tex = torch.nn.Parameter(torch.full((R, R, 3), 0.5, device=self.dev))
            self.ks_tex = torch.nn.Parameter((lo + ks0 * (hi - lo)).to(self.dev))
            self._nrm_min = torch.tensor([-1.0, -1.0, 0.0], device=self.dev)           # FLAGS.nrm_min / nrm_max (train.py:552-553)
            self._nrm_max = torch.tensor([1.0, 1.0, 1.0], device=self.dev)
            self.params = [self.kd_tex, self.ks_tex]
            names = ['kd', 'ks']
            clamps = [(0.0, 1.0), (None, None, self._ks_min, self._ks_max)]
            lr_scales, grad_scales, norm3 = [1.0, 1.0], [1.0, 1.0], [False, False]
            self.nrm_tex = None
            if self.perturbed_nrm:
                self.nrm_tex = torch.nn.Parameter(torch.tensor([0.0, 0.0, 1.0], device=self.dev).repeat(R, R, 1).contiguous())
                self.params.append(self.nrm_tex)
                names.append('normal')
                clamps.append((None, None, self._nrm_min, self._nrm_max)); lr_scales.append(1.0); grad_scales.append(1.0); norm3.append(True)
            self.params.append(self.light.base)
            names.append('light')
            clamps.append((0.01, None)); lr_scales.append(lr_light / lr); grad_scales.append(light_grad_scale); norm3.append(False)
            if self.optimize_geometry:
                # DLMesh: v_pos is the trained geometry (dlmesh.py:28-38); the target was rendered from the unperturbed mesh above
                v0 = self.mesh['v_pos'].clone()
                if perturb_pos:
                    v0 = v0 + perturb_pos * torch.randn(v0.shape, generator=torch.Generator().manual_seed(77)).to(self.dev)
                self.v_pos = torch.nn.Parameter(v0.contiguous())
                self.params.append(self.v_pos)
                names.append('v_pos')
                clamps.append(None); lr_scales.append(lr_pos / lr); grad_scales.append(1.0); norm3.append(False)
        self.param_names = names
        self.n_tex = sum(1 for nm in names if nm in ('kd', 'ks', 'normal')) if material_set == 'full' else 0     # the leading texture parameters
        # The same Adam as the reference (train.py:348-356,452-461: three optimizers that differ in their learning rate).  fused: ONE
        # launch for the light-gradient scale, the Adam update of every tensor and the clamps / normal-map renormalisation
        # (csrc/optim.hip; torch's multi-tensor Adam puts the elements on 16 workgroups + a kernel per clamp); otherwise
        # torch.optim.Adam with parameter groups and the reference's sequence of calls.
        self._fused_update = bool(fused) and self.dev.type == 'cuda'
        self._lr_scales = lr_scales
        self._tex_grad = None
        if self._fused_update:
            sparse = [nm in ('kd', 'ks', 'normal') and material_set == 'full' for nm in names]
            self.opt = FusedAdam(self.params, lr=lr, grad_scales=grad_scales, clamps=clamps, lr_scales=lr_scales, normalize3=norm3,
                                 sparse=sparse, zero_grad=sparse)
            if material_set == 'full':
                self._tex_grad = rd.PersistentGrads(torch.zeros_like(p) for p in self.params[:self.n_tex])
        else:
            groups = [{'params': [p], 'lr': lr * sc_} for p, sc_ in zip(self.params, lr_scales)]
            try:
                self.opt = torch.optim.Adam(groups, lr=lr, fused=self.dev.type == 'cuda', capturable=use_graph)
            except (RuntimeError, TypeError):
                self.opt = torch.optim.Adam(groups, lr=lr, capturable=use_graph)
        self.covered = int(self.mask.sum().item())
        # Where the per-iteration BVH rebuild runs (OptiXContext.set_build_mode): 1 = the context's side stream (default).  (Mode 2 -- the
        # build's launches issued behind the sample generation's -- was measured on the one-view iteration in HIP graphs: 2.021 vs 2.018 ms, no
        # difference; a replayed graph orders its nodes by their dependencies, not by the order they were captured in.)
        mode = self._build_mode
        if mode is None:
            env = _lib.tuning_env('NVDR_BUILD_MODE')
            mode = int(env) if env is not None else 1
        if mode != 1:
            self.ctx.check()             # (the construction-time build and G-buffer are done: no build in flight)
            self.ctx.set_build_mode(mode)

    def _set_gbuffer(self, gb):
        """Adopt a G-buffer dict (optixutils.render_gbuffer / render.gbuffer) as the iteration's inputs."""
        self.rast = gb['rast']
        self.mask = (gb['rast'][..., 3] > 0).float().contiguous()     # the reference passes rast[..., -1] (triangle id + 1 > 0)
        self.gb_pos = gb['gb_pos']
        self.gb_geom_nrm = gb['gb_geometric_normal']
        self.gb_smooth_nrm = gb['gb_normal']
        self.gb_tangent = gb['gb_tangent']
        self.gb_depth = gb['gb_depth']
        self.gb_texc = gb['gb_texc']

    @property
    def seed(self):
        return int(self.seed_dev.item())

    @seed.setter
    def seed(self, v):
        self.seed_dev.fill_(int(v))

    # rays per pass counted from the actual mask: 2 per stratum per covered pixel
    def rays_per_pass(self):
        return 2 * self.n * self.n * self.covered

    @torch.no_grad()
    def shade_inputs(self):
        """(mask, ro, gb_pos, shading normal, view_pos, kd, ks) for the CURRENT parameters: what optix_env_shade receives inside the
        iteration (counting launches of bench.py, the probes under tools/)."""
        if self.material_set == 'r3':
            m = self.mask[..., None]
            kd = self.kd_tex[self.texel].view(self.nv, self.res, self.res, 3) * m
            ks = self.ks.view(1, 1, 1, 3) * m
            pn = None
        else:
            kd, ks, pn = self._lookup(self.kd_tex, self.ks_tex, self.nrm_tex, self.gb_texc, self.rast)
        nrm = ru.prepare_shading_normal(self.gb_pos, self.view_pos, pn, self.gb_smooth_nrm, self.gb_tangent, self.gb_geom_nrm,
                                        two_sided_shading=True, opengl=True)
        return self.mask, (self.gb_pos + nrm * 0.001).contiguous(), self.gb_pos, nrm, self.view_pos, kd.contiguous(), ks.contiguous()

    def _render(self, kd_tex, ks_vec, light):
        m = self.mask[..., None]
        if self.fused and self.dev.type == 'cuda':
            kd = _lookup_rows.apply(kd_tex, self.texel_or_none)
        else:
            kd = torch.zeros(self.nv * self.res * self.res, 3, device=self.dev).index_copy(0, self.cov, _gather_rows.apply(kd_tex, self.texel_cov))
        kd = kd.view(self.nv, self.res, self.res, 3)
        # uncovered pixels are skipped by the mask in env-shade and have zero light in the composite: no ks * mask needed
        ks = _broadcast_pixels.apply(ks_vec, self.nv, self.res, self.res) if self.fused else (ks_vec.view(1, 1, 1, 3) * m)
        nn = None
        if self.fused:
            # shading normal, its unit copy (the filter's guide) and the shadow-ray origin in one launch (eight otherwise)
            nrm, nn, ro = ru.shading_frame(self.gb_pos, self.view_pos, None, self.gb_smooth_nrm, self.gb_tangent, self.gb_geom_nrm,
                                           two_sided_shading=True, opengl=True, ro_eps=0.001)
        else:
            nrm = ru.prepare_shading_normal(self.gb_pos, self.view_pos, None, self.gb_smooth_nrm, self.gb_tangent,
                                            self.gb_geom_nrm, two_sided_shading=True, opengl=True)
            ro = self.gb_pos + nrm * 0.001
        self.ctx.pixel_index_offset = self.pixel_index_offset          # per-context switches (ops.OptiXContext)
        self.ctx.cache_visibility = not self.retrace_backward
        diff, spec = ou.optix_env_shade(self.ctx, self.mask, ro, self.gb_pos, nrm, self.view_pos, kd, ks, light.base,
                                        light._pdf, light.rows[:, 0], light.cols, BSDF='pbr', n_samples_x=self.n,
                                        rnd_seed=0, shadow_scale=1.0)      # effective seed = 0 + the device counter
        if self.denoiser is not None and not self.denoiser_demodulate:
            # the non-demodulated branch of shade() (render.py:124-131): ONE filter pass over the combined colour
            shaded = ru.shade_composite(diff, spec, kd, ks) if self.fused else diff * (kd * (1.0 - ks[..., 2:3])) + spec
            if self.fused:
                cw = ou.ops._bilateral_denoiser_func.apply(shaded, nn if nn is not None else _safe_normalize(nrm), self.gb_depth, self.denoiser.sigma)
                return cw[..., 0:3] / cw[..., 3:4]
            return self.denoiser.forward(torch.cat((shaded, nrm, self.gb_depth), dim=-1))
        if self.fused:
            # same arithmetic as the branch below; the normal is normalised once for both filter passes, the filter
            # kernel is called without the 8-channel cat, and its (colour sum, weight) output goes straight into the
            # fused composite: ~45 small torch kernels per iteration less
            if self.denoiser is not None:
                if nn is None:
                    nn = _safe_normalize(nrm)
                if self.pair_filter:     # both images in one pass: same guides, same weights (bit-identical to the two calls below)
                    diff, spec = ou.ops._bilateral_denoiser_pair_func.apply(diff, spec, nn, self.gb_depth, self.denoiser.sigma)
                else:
                    diff = ou.ops._bilateral_denoiser_func.apply(diff, nn, self.gb_depth, self.denoiser.sigma)
                    spec = ou.ops._bilateral_denoiser_func.apply(spec, nn, self.gb_depth, self.denoiser.sigma)
            return ru.shade_composite(diff, spec, kd, ks)
        if self.denoiser is not None:   # the reference's own sequence of calls (render.py:119-127)
            diff = self.denoiser.forward(torch.cat((diff, nrm, self.gb_depth), dim=-1))
            spec = self.denoiser.forward(torch.cat((spec, nrm, self.gb_depth), dim=-1))
        return diff * (kd * (1.0 - ks[..., 2:3])) + spec

    @staticmethod
    def _lookup(kd_tex, ks_tex, nrm_tex, texc, rast, grad_buffers=None):
        """kd, ks and the perturbed normal (None without a normal map: no_perturbed_nrm) of every pixel in one launch."""
        if nrm_tex is None:
            kd, ks = rd.texture_lookup((kd_tex, ks_tex), texc, rast, grad_buffers=grad_buffers)
            return kd, ks, None
        return rd.texture_lookup((kd_tex, ks_tex, nrm_tex), texc, rast, grad_buffers=grad_buffers)

    def _render_full(self, kd_tex, ks_tex, nrm_tex, light, gb=None, grad_buffers=None, loss_against=None):
        """shade() with the reference's material set (render.py:61-131): kd / ks / perturbed normal from three textures in one
        lookup launch, shading frame, env-shade, both lights filtered in one pass, composite.  gb: a differentiable G-buffer dict
        (optimize_geometry) or None = the cached one."""
        rast = self.rast if gb is None else gb['rast']
        pos, gnrm, snrm, tng, depth, texc = ((self.gb_pos, self.gb_geom_nrm, self.gb_smooth_nrm, self.gb_tangent, self.gb_depth, self.gb_texc) if gb is None else
                                             (gb['gb_pos'], gb['gb_geometric_normal'], gb['gb_normal'], gb['gb_tangent'], gb['gb_depth'], gb['gb_texc']))
        self.ctx.pixel_index_offset = self.pixel_index_offset
        self.ctx.cache_visibility = not self.retrace_backward
        if not (self.fused and self.dev.type == 'cuda'):
            # the same iteration written as the reference writes it (render.py:61-131): torch indexing for the texel lookups, the
            # operator API for the rest, the 8-channel cat in front of each filter pass, the composite in torch
            m = (rast[..., 3:4] > 0).float()

            def look(t):
                R = t.shape[0]
                ix = (texc[..., 0] * R).long().clamp(0, R - 1)
                iy = ((1.0 - texc[..., 1]) * R).long().clamp(0, R - 1)
                return t[iy, ix] * m
            kd, ks, pn = look(kd_tex), look(ks_tex), (look(nrm_tex) if nrm_tex is not None else None)
            nrm = ru.prepare_shading_normal(pos, self.view_pos, pn, snrm, tng, gnrm, two_sided_shading=True, opengl=True)
            ro = pos + nrm * 0.001
            diff, spec = ou.optix_env_shade(self.ctx, rast[..., 3], ro, pos, nrm, self.view_pos, kd, ks, light.base,
                                            light._pdf, light.rows[:, 0], light.cols, BSDF='pbr', n_samples_x=self.n,
                                            rnd_seed=0, shadow_scale=1.0)
            if self.denoiser is not None and not self.denoiser_demodulate:
                return self.denoiser.forward(torch.cat((diff * (kd * (1.0 - ks[..., 2:3])) + spec, nrm, depth), dim=-1))
            if self.denoiser is not None:
                diff = self.denoiser.forward(torch.cat((diff, nrm, depth), dim=-1))
                spec = self.denoiser.forward(torch.cat((spec, nrm, depth), dim=-1))
            return diff * (kd * (1.0 - ks[..., 2:3])) + spec
        kd, ks, pn = self._lookup(kd_tex, ks_tex, nrm_tex, texc, rast, grad_buffers=grad_buffers)
        nrm, nn, ro = ru.shading_frame(pos, self.view_pos, pn, snrm, tng, gnrm, two_sided_shading=True, opengl=True, ro_eps=0.001)
        self.ctx.cache_visibility = not self.retrace_backward
        # mask = rast[..., -1] as the reference passes it (render.py:113): a strided view, > 0 = covered
        diff, spec = ou.optix_env_shade(self.ctx, rast[..., 3], ro, pos, nrm, self.view_pos, kd, ks, light.base,
                                        light._pdf, light.rows[:, 0], light.cols, BSDF='pbr', n_samples_x=self.n,
                                        rnd_seed=0, shadow_scale=1.0)
        if self.denoiser is not None and not self.denoiser_demodulate:
            shaded = ru.shade_composite(diff, spec, kd, ks)
            cw = ou.ops._bilateral_denoiser_func.apply(shaded, nn, depth, self.denoiser.sigma)
            return cw[..., 0:3] / cw[..., 3:4]
        if self.denoiser is not None:
            if self.pair_filter:
                diff, spec = ou.ops._bilateral_denoiser_pair_func.apply(diff, spec, nn, depth, self.denoiser.sigma)
            else:
                diff = ou.ops._bilateral_denoiser_func.apply(diff, nn, depth, self.denoiser.sigma)
                spec = ou.ops._bilateral_denoiser_func.apply(spec, nn, depth, self.denoiser.sigma)
        if loss_against is not None:
            # the composite, the mean image loss and both adjoints in ONE launch (round 6: five small launches and their dependency gaps less per
            # iteration; same values bit for bit).  loss_against = (target, upstream gradient of the mean: the resident scalar backward() is called with)
            return ru.shade_composite_loss(diff, spec, kd, ks, loss_against[0], loss_against[1], loss='l1', tonemapper='log_srgb')
        return ru.shade_composite(diff, spec, kd, ks)

    def _refit_now(self, ahead=0):
        """Iteration _iter + ahead refits instead of rebuilding (rebuild_every; trained geometry only)."""
        return self.optimize_geometry and self.rebuild_every > 1 and (self._iter + ahead) % self.rebuild_every != 0

    def _build_bvh(self, rebuild=None, ahead=0):
        v_pos = self.v_pos if self.optimize_geometry else self.mesh['v_pos']
        if rebuild is None:
            rebuild = 0 if self._refit_now(ahead) else 1
        ou.optix_build_bvh(self.ctx, v_pos, self.mesh['t_pos_idx'], rebuild=rebuild)
        self._build_deferred = False
        return v_pos

    def _stage1(self, defer_build=False, build=True, ahead=0):
        """The geometry stage of an iteration -- everything that needs only the light probe and the vertices, not the textures: BVH
        rebuild (side stream), update_pdf, and with trained geometry getMesh (dlmesh.py:45-55) + rasterize / interpolate
        (render.py:208-234) from the moving vertices.  With several ranks it runs while the texture chunk of the previous iteration's
        gradient exchange is still on the wire (_step_multi).
        defer_build (a HIP graph of this stage alone, trained geometry or the unpipelined schedule): the stage ends with the main stream
        waiting for the rebuild -- a graph must not end with the side stream's work unjoined.  With LOCKED geometry under the pipelined
        schedule nothing of stage 2 but the traversal needs the tree, and the stage is cut differently (round 6, _capture): the rebuild is
        a graph of its own replayed on a side stream as soon as the early chunks are updated -- it runs while the texture chunk is on the
        wire AND, if it is not done by then, beside the lookups and the sample generation like in the one-rank iteration; the main stream
        waits for it in front of the traversal kernel only."""
        # the rebuild first: it runs on the context's side stream, and the sooner it starts the less of it is left when the traversal
        # needs the tree (one view: the light's three small kernels used to run in front of it)
        self._build_deferred = False
        if build:
            v_pos = self._build_bvh(ahead=ahead)       # (ahead = 1: the pipelined geometry stage at the end of step i belongs to iteration i + 1)
        else:       # (the rebuild / refit has been replayed as a graph of its own in front of this stage: _capture, rebuild_every)
            v_pos = self.v_pos if self.optimize_geometry else self.mesh['v_pos']
        self.light.update_pdf()
        if defer_build and not self.optimize_geometry:
            self.ctx.wait_build()            # (only the un-split HIP graph of this stage: it must not end with the side stream unjoined)
        self._gb_live = None
        if self.material_set != 'r3' and self.optimize_geometry:
            v_nrm, v_tng = mesh_ops.mesh_frame(v_pos, self.topo)
            gb = rd.gbuffer(self.ctx, v_pos, v_nrm, v_tng, self.topo, self.mvp, self.cam, (self.res, self.res))
            self._set_gbuffer({k: v.detach() for k, v in gb.items()})        # shade_inputs() / mask follow the moving mesh
            self._gb_live = gb
        self._stage1_ready = True

    def _stage2(self, grad_seed=None):
        """Texture lookups, shading frame, env-shade, filters, composite, loss and the whole backward pass; returns the loss tensor
        (gradients are in .grad, the texture gradients scatter-added into the persistent buffers / exchange buckets)."""
        self._stage1_ready = False
        if getattr(self, '_build_deferred', False):
            self._build_bvh()
        self.opt.zero_grad(set_to_none=True)
        if self._tex_grad is not None and self._tex_grad.dirty:
            self._zero_tex_grad()           # a backward pass whose gradients no update consumed (forward_backward called on its own)
        if getattr(self, '_one', None) is None:
            self._one = torch.ones((), dtype=torch.float32, device=self.dev)
        fuse_loss = (self.fused_loss and self.fused and self.dev.type == 'cuda' and self.material_set != 'r3'
                     and (self.denoiser is None or self.denoiser_demodulate))
        if self.material_set == 'r3':
            img = self._render(self.kd_tex, self.ks, self.light)
        else:
            img = self._render_full(self.kd_tex, self.ks_tex, self.nrm_tex, self.light, self._gb_live, grad_buffers=self._tex_grad if self.fused else None,
                                    loss_against=(self.target, self._one) if fuse_loss else None)
        if fuse_loss:
            loss = img
        else:
            loss = (ru.image_loss_mean if (self.fused and self.dev.type == 'cuda') else ru.image_loss)(img, self.target, loss='l1', tonemapper='log_srgb')
        if self._one.shape != loss.shape or self._one.device != loss.device:
            self._one = torch.ones_like(loss)
        loss.backward(gradient=self._one)       # (a resident seed: no fill launch per iteration)
        self._gb_live = None
        return loss

    def forward_backward(self):
        """The differentiable part of the iteration; returns the loss tensor (grads are in .grad)."""
        if not getattr(self, '_stage1_ready', False):
            self._stage1()
        return self._stage2()

    def set_lr_scale(self, name, value):
        """Learning rate of one parameter tensor relative to lr (0 freezes it); names as in .param_names.  Completes a pending
        (pipelined) texture update first, so that the new rate applies from the next iteration on."""
        self.finish()
        i = self.param_names.index(name)
        self._lr_scales[i] = float(value)
        if self._fused_update:
            self.opt.lr_scales[i] = float(value)
        else:
            self.opt.param_groups[i]['lr'] = self.opt.defaults['lr'] * float(value)

    def _chunk_indices(self):
        """The gradient exchange's chunks as lists of parameter indices, in the order the NEXT iteration needs them: [light, v_pos] --
        update_pdf, BVH build, vertex frames and G-buffer wait for these -- then the textures [kd, ks, normal] (25-38 MB at 1024^2), which
        only the texture lookup waits for.  One chunk for round 3's small set."""
        n = len(self.params)
        if self.material_set == 'r3' or self.n_tex == 0 or self.n_tex == n:
            return [list(range(n))]
        return [list(range(self.n_tex, n)), list(range(self.n_tex))]

    def _update(self, subset=None, advance=True, grad_mult=1.0):
        """Everything after the gradient exchange: light-gradient scale, Adam, clamps (train.py:439-476); subset = the parameter
        indices of one exchange chunk (fused path only)."""
        if self._fused_update:
            self.opt.step(subset=subset, advance=advance, grad_mult=grad_mult)
            return
        if grad_mult != 1.0:
            for p in self.params:
                if p.grad is not None:
                    p.grad.mul_(grad_mult)
        if self.light.base.grad is not None and self.light_grad_scale != 1.0:
            self.light.base.grad *= self.light_grad_scale       # train.py:439-440
        self.opt.step()
        with torch.no_grad():
            self.kd_tex.clamp_(0.0, 1.0)
            if self.material_set == 'r3':
                self.ks.copy_(torch.maximum(self.ks.clamp(max=1.0), self._ks_min))  # ks_min of configs/bob.json: roughness >= 0.08
                self.light.base.clamp_(min=0.0)
            else:                                       # train.py:467-476: Texture2D.clamp_ per channel, normalize_, lgt.clamp_(min=0.01)
                for i in range(3):
                    self.ks_tex[..., i].clamp_(min=float(self._ks_min[i]), max=float(self._ks_max[i]))
                    if self.nrm_tex is not None:
                        self.nrm_tex[..., i].clamp_(min=float(self._nrm_min[i]), max=float(self._nrm_max[i]))
                if self.nrm_tex is not None:
                    self.nrm_tex.copy_(self.nrm_tex / torch.sqrt(torch.clamp((self.nrm_tex * self.nrm_tex).sum(-1, keepdim=True), min=1e-20)))
                self.light.base.clamp_(min=0.01)

    def _exchange(self, world_size):
        """The chunked gradient exchange of this step object (created on first use: the process group must exist by then)."""
        if getattr(self, '_ex', None) is None or self._ex.world != world_size:
            from .parallel import GradientExchange
            total = self.total_views
            # (one rank under force_exchange: its views are the whole batch as far as the weighting goes)
            even = world_size == 1 or ((total % world_size == 0) and (self.nv * world_size == total))
            chunks = self._chunk_indices() if self._fused_update else [list(range(len(self.params)))]
            groups = [[self.params[i] for i in idx] for idx in chunks]
            # tile-sparse: the texture chunk only ('dense' sends the whole bucket; 'sparse' / 'auto' send the touched tiles and fall back to
            # the dense bucket when more than half of the tiles are touched -- with the reference's mip-mapped textures they all would be)
            pol = {'dense': False, 'sparse': True, 'auto': 'auto'}[self.exchange_mode]
            sparse = [pol if (len(chunks) == 2 and k == 1) else False for k in range(len(chunks))]
            self._ex = GradientExchange(groups, world_size, local_weight=self.nv, equal_shards=even, sparse=sparse)
            self._ex_chunks = chunks if self._fused_update else [None]
            # Several ranks: the texture lookup's adjoint scatter-adds straight into the exchange buckets (they are what the persistent
            # gradient buffers were: all zero between iterations, re-zeroed tile by tile by the optimizer that consumes them), so the
            # 37.7 MB of texture gradients are neither packed nor cleared: one copy and three memsets less per iteration and rank.
            self._tex_grad_resident = False
            if self._tex_grad is not None and (world_size > 1 or self.force_exchange):
                self._tex_grad = rd.PersistentGrads(self._ex.slot(p) for p in self.params[:self.n_tex])
                self._tex_grad_resident = True
            self._pending = False
            if self._union_views and any(self._ex.sparse):
                self._ex.extra_flags[self._ex.sparse.index(True)] = self._union_flags()
        return self._ex

    @torch.no_grad()
    def _union_flags(self):
        """One byte per 64-texel tile of the texture chunk: 1 where some covered pixel of the union views looks a texel of that tile up."""
        from .parallel import TILE_FLOATS
        cams = [sc.camera(vw, self.total_views) for vw in self._union_views]
        mvp = torch.stack([c[1] for c in cams]).to(self.dev)
        cam = torch.stack([sc.camera_rays(c[0]) for c in cams]).to(self.dev)
        gb = ou.render_gbuffer(self.ctx, self.mesh, mvp, cam, (self.res, self.res))
        cov = gb['rast'][..., 3].reshape(-1) > 0
        tc = gb['gb_texc'].reshape(-1, 2)[cov]
        out = []
        for p in self.params[:self.n_tex]:
            R = p.shape[0]
            ix = (tc[:, 0] * R).long().clamp(0, R - 1)
            iy = ((1.0 - tc[:, 1]) * R).long().clamp(0, R - 1)
            f = torch.zeros(p.numel() // TILE_FLOATS, dtype=torch.uint8, device=self.dev)
            f[(iy * R + ix) // (TILE_FLOATS // 3)] = 1
            out.append(f)
        return torch.cat(out)

    # ---- several ranks (or force_exchange): the exchange is pipelined with the next iteration ---------------------------------------
    #
    #   step i:      stage 2 (lookups ... backward, pack, tile flags) | all-reduce [light, v_pos] ; all-reduce tile flags
    #                Adam [light, v_pos]  ->  stage 1 of step i + 1 (BVH rebuild, update_pdf, vertex frames, G-buffer)     <- main stream
    #                texture chunk: union list, gather, all-reduce                                                         <- side stream + RCCL
    #   step i + 1:  wait(textures) -> scatter -> Adam [kd, ks, normal] -> stage 2 ...
    #
    # The textures' update completes before the next texture lookup, so every iteration sees exactly the parameters of the unpipelined
    # schedule; finish() completes the last one.

    def _wait(self, k):
        """ex.wait(k) with the time the MAIN stream stands still for it measured by a pair of events (measure_exposed)."""
        ex = self._ex
        if not self.measure_exposed:
            return ex.wait(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f = ex.wait(k)
        b.record()
        self._exposed_events.append((a, b))
        return f

    def exposed_ms(self):
        """Per-iteration time the main stream waited on the exchange (collectives + scatter) since measure_exposed was set, as a list; clears it."""
        torch.cuda.synchronize()
        per_wait = [a.elapsed_time(b) for a, b in self._exposed_events]
        self._exposed_events = []
        n = max(1, len(self._ex_chunks))
        return [sum(per_wait[i:i + n]) for i in range(0, len(per_wait) - n + 1, n)]

    def stage1_ms(self):
        """Durations of the pipelined geometry stage on the main stream since measure_exposed was set (a list; clears it)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._stage1_events]
        self._stage1_events = []
        return out

    def _finish_pending(self, graphs=None):
        """The texture chunk of the previous iteration: wait for its all-reduce, scatter, Adam + clamps."""
        if not getattr(self, '_pending', False):
            return
        k = len(self._ex_chunks) - 1
        f = self._wait(k)
        if graphs is not None:
            graphs[k].replay()
        else:
            self._update(subset=self._ex_chunks[k], advance=True, grad_mult=f)
        self._pending = False

    def parameters(self):
        """{name: tensor} of the trained set in a CONSISTENT state: with several ranks step() returns with the texture chunk's update
        still pending (see step()); this completes it first.  Use it (or call finish()) before a checkpoint, a validation render or any
        other read of .params between steps."""
        self.finish()
        return dict(zip(self.param_names, self.params))

    def __del__(self):
        try:
            if getattr(self, '_pending', False):
                import warnings
                warnings.warn('DirectLightingStep dropped with the last texture update still pending: call finish() after the last step()')
        except Exception:
            pass

    def finish(self):
        """Complete the parameter update of the last iteration (the pipelined texture chunk).  Call once after the last step()."""
        if getattr(self, '_pending', False):
            g = self._graphs[1] if (self._graphs is not None and not self.force_eager) else None
            self._finish_pending(g)
        self._join_build_side()

    def _step_multi(self, world_size):
        ex = self._exchange(world_size)
        chunks = self._ex_chunks
        graphs = self._graphs if (self._graphs is not None and not self.force_eager) else None
        if graphs is None and self._graphs is not None and not getattr(self, '_left_graphs', False):
            self._left_graphs, self._stage1_ready = True, False        # force_eager: the geometry stage is redone eagerly (its autograd graph too)
            self._join_build_side()
        last = len(chunks) - 1
        self._finish_pending(graphs[1] if graphs else None)
        split = graphs is not None and isinstance(graphs[2], tuple)      # (G2a, G2b): the stage-2 graph cut in front of the traversal
        if graphs:
            g1, gb, g2 = graphs[0], graphs[1], graphs[2]
            if not self._stage1_ready:
                self._replay_stage1(g1, split, 0)
                self._stage1_ready = True
            if split:
                g2[0].replay()                                           # update_pdf, lookups, shading frame, sample generation
                torch.cuda.current_stream(self.dev).wait_event(self._ev_build)
                g2[1].replay()                                           # traversal ... backward, bucket pack
            else:
                g2.replay()
            self._stage1_ready = False
            loss = self._loss_static
            ex.compute_flags()          # (not captured: whether this round goes through the tiles is the exchange's decision, round by round)
        else:
            self._eager_steps += 1
            if not self._stage1_ready:
                self._stage1(defer_build=self.optimize_geometry or not self.pipeline)
            loss = self._stage2()
            ex.pack()
            self._packed_tex_grad()
            ex.compute_flags()
        ex.start(skip_single=not self.force_exchange)
        for k in range(last):                       # the early chunks: their update, then the next iteration's geometry stage can go
            f = self._wait(k)
            if graphs:
                gb[k].replay()
            else:
                self._update(subset=chunks[k], advance=False, grad_mult=f)
        if self.pipeline and last > 0:
            if self.measure_exposed:        # how long the geometry stage keeps the main stream busy: what the texture chunk's wire time hides behind
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            if graphs:
                self._replay_stage1(graphs[0], split, 1)
                self._stage1_ready = True
            else:
                self._stage1(defer_build=self.optimize_geometry, ahead=1)
            if self.measure_exposed:
                b.record()
                self._stage1_events.append((a, b))
        ex.send(last)                               # sparse: the host reads the union's size here -- the GPU is busy with stage 1 meanwhile
        self._pending = True
        if not (self.pipeline and last > 0):
            self._finish_pending(graphs[1] if graphs else None)
        self.allreduce_bytes = ex.bytes_per_step
        return loss

    def _join_build_side(self):
        """Split schedule: a rebuild replayed on the side stream may still be running with nobody but the next G2b waiting for it; whoever
        touches the tree or rebuilds it outside the graphs (finish(), leaving graph mode) orders the main stream behind it first."""
        if getattr(self, '_ev_build', None) is not None and self._graphs is not None and isinstance(self._graphs[2], tuple):
            torch.cuda.current_stream(self.dev).wait_event(self._ev_build)

    def _replay_stage1(self, g1, split, ahead=0):
        """The geometry stage of the NEXT iteration.  Un-split: on the main stream (trained geometry: everything behind it needs its
        G-buffer).  Split (locked geometry): the rebuild alone, on a side stream ordered behind this point of the main stream (the last
        reader of the old tree -- the backward pass's traversal -- lies before it); the main stream moves on and meets it again in front
        of the next traversal (_ev_build)."""
        if not split:
            gv = getattr(self, '_gb_variants', None)
            if gv is not None:
                # (ahead = 1: the geometry stage replayed at the end of step i belongs to iteration i + 1)
                gv[1 if self._refit_now(ahead) else 0].replay()
            g1.replay()
            return
        main = torch.cuda.current_stream(self.dev)
        self._ev_main.record(main)
        self._build_side.wait_event(self._ev_main)
        with torch.cuda.stream(self._build_side):
            g1.replay()
            self._ev_build.record(self._build_side)

    def _tex_grad_guard(self):
        """One rank: FusedAdam zeroes the texture gradients it consumed, which clears the persistent scatter-add buffers only if
        p.grad IS that buffer.  Should autograd have copied instead of adopting (a hook, a second reference), clear them explicitly."""
        if self._tex_grad is not None:
            self._tex_grad.dirty = False
            for p, b in zip(self.params[:self.n_tex], self._tex_grad):
                if p.grad is None or p.grad.data_ptr() != b.data_ptr():
                    self._zero_tex_grad()
                    return

    def _packed_tex_grad(self):
        """The gradients are in the exchange buckets: clear the persistent scatter-add buffers of the texture lookup -- unless they ARE
        the buckets (_exchange), which the optimizer re-zeroes as it consumes them."""
        if getattr(self, '_tex_grad_resident', False):
            self._tex_grad.dirty = False
        else:
            self._zero_tex_grad()

    def _zero_tex_grad(self):
        """Several ranks: the optimizer reads (and zeroes) the exchange buckets, not the persistent scatter-add buffers of the texture
        lookup -- those are cleared here, once their content has been packed."""
        if self._tex_grad is not None:
            self._tex_grad.dirty = False
            for b in self._tex_grad:
                b.zero_()

    def _capture(self, world_size):
        """HIP graphs.  One rank: ONE graph -- update_pdf + BVH rebuild + render + loss + backward + light-gradient scale + Adam + clamps.
        Several ranks: G1 (stage 1: the geometry stage), G2 (stage 2 + bucket pack + tile flags), B_k (Adam + clamps of exchange chunk k);
        the collectives run between the replays, on RCCL's stream."""
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)
        # The parameters' gradient accumulators were created on the default stream by the eager iterations; the capture runs on torch's
        # capture stream.  Every .grad is None here, so an accumulator only ADOPTS the incoming tensor -- it launches nothing that the
        # capture could miss -- and torch's warning about the stream mismatch does not apply.
        warn = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if warn is not None:
            warn(False)
        multi = world_size > 1 or self.force_exchange
        if not multi:
            if self.optimize_geometry and self.rebuild_every > 1:
                # the whole iteration twice: with a rebuild and with a refit of the tree (each graph has its own pool: they never run together)
                gas, losses = [], []
                it = self._iter
                for k in (0, 1):
                    self._iter = k              # (_build_bvh reads the flag from the iteration counter)
                    g = torch.cuda.CUDAGraph()
                    with _capture_graph(g):
                        self._stage1_ready = False
                        losses.append(self.forward_backward())
                        self._update()
                        self._tex_grad_guard()
                    gas.append(g)
                    self.opt.zero_grad(set_to_none=True)
                self._iter = it
                self._loss_static = losses
                self._graphs = (tuple(gas), None, None)
                return
            ga = torch.cuda.CUDAGraph()
            with _capture_graph(ga):
                self._stage1_ready = False
                self._loss_static = self.forward_backward()
                self._update()
                self._tex_grad_guard()
            self._graphs = (ga, None, None)
            return
        ex = self._exchange(world_size)
        g1 = torch.cuda.CUDAGraph()
        # (not with a tile-sparse chunk: its send() blocks the host on a side stream in the middle of the schedule, and together with the rebuild's
        # side stream that combination stalled a queue in round 6 -- session 10; the tile-sparse exchange keeps the un-split graphs)
        split = bool(self.pipeline and not self.optimize_geometry and len(self._ex_chunks) > 1 and self.material_set == 'full'
                     and not any(ex.sparse) and _lib.tuning_env('NVDR_SPLIT_STAGE2', '1') != '0')
        if split:
            # LOCKED geometry: G1 = the rebuild alone (joined inside the graph), replayed on a side stream; stage 2 = TWO graphs cut between
            # the sample generation and the traversal (OptiXContext.split_hook: the env-shade launch is issued in two calls), so that the
            # main stream waits for the tree where the one-rank iteration does.  (An external-event wait node inside ONE graph would do the
            # same; this runtime does not capture them: torch refuses external events on ROCm, and hipStreamWaitEvent(.., hipEventWaitExternal)
            # on a capturing stream joins the recording stream into the capture instead -- "capturing stream has unjoined work".)
            if getattr(self, '_build_side', None) is None:
                # HIGH priority: HIP deals its streams to four hardware queues per priority level, and a normal-priority stream created
                # late in a process may share the main stream's queue -- the first measurement of this schedule showed the main stream
                # standing still for the whole rebuild (tools/side_graph_probe.py: one in four fresh normal streams serialises with the
                # main stream, none of the high-priority ones does); the rebuild is on the traversal's critical path anyway
                self._build_side = torch.cuda.Stream(device=self.dev, priority=-1)
                self._ev_build, self._ev_main = torch.cuda.Event(), torch.cuda.Event()
            with _capture_graph(g1):
                self._build_bvh()
                self.ctx.wait_build()
            self.ctx.build_joined()              # replays are ordered by _ev_build: no consumer waits on the build's own event
            g2a, g2b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            pool2 = torch.cuda.graph_pool_handle()       # (not G1's: that graph replays on another stream, concurrently with these two)
            cm_a, cm_b = _capture_graph(g2a, pool=pool2), _capture_graph(g2b, pool=pool2)
            cut = []

            def hook():
                cm_a.__exit__(None, None, None)
                cut.append(True)
                cm_b.__enter__()
            self.ctx.split_hook = hook
            cm_a.__enter__()
            try:
                self.light.update_pdf()
                self._gb_live, self._stage1_ready = None, True
                self._loss_static = self._stage2()
                ex.pack()
                self._packed_tex_grad()
            finally:
                self.ctx.split_hook = None
                (cm_b if cut else cm_a).__exit__(None, None, None)
            if not cut:
                raise RuntimeError('the env-shade launch did not reach its split point')
            g2 = (g2a, g2b)
            self.split_stage2 = True
        else:
            self._gb_variants = None
            if self.optimize_geometry and self.rebuild_every > 1:
                # trained geometry with a refit policy: the tree's graph comes in two variants (rebuild, refit) replayed in front of the rest
                # of the geometry stage -- nothing in them belongs to torch's allocator, so either may precede the same G1
                gv = []
                for rb in (1, 0):
                    g = torch.cuda.CUDAGraph()
                    with _capture_graph(g):
                        self._build_bvh(rebuild=rb)
                        self.ctx.wait_build()
                    gv.append(g)
                self._gb_variants = tuple(gv)
            with _capture_graph(g1):
                self._stage1(defer_build=True, build=self._gb_variants is None)
            g2 = torch.cuda.CUDAGraph()
            with _capture_graph(g2, pool=g1.pool()):
                self._loss_static = self._stage2()
                ex.pack()
                self._packed_tex_grad()
        gbs = []
        n = len(self._ex_chunks)
        # The B_k graphs get a pool of their own: the pipelined replay order (G2, B_0, G1, B_1) differs from the capture order, and with
        # trained geometry the tensors G1's autograd graph saved must survive from G1 to G2's backward -- B_1 replays in between and must
        # not be handed their memory.  (The fused update allocates nothing today: asserted below, so that a change of it is noticed.)
        pool_b = None
        for k in range(n):
            f = self._point_grads(k)        # the .grad of chunk k = views into its bucket: what the captured update reads on every replay
            gk = torch.cuda.CUDAGraph()
            before = torch.cuda.memory_allocated(self.dev)
            with _capture_graph(gk, **({'pool': pool_b} if pool_b is not None else {})):
                self._update(subset=self._ex_chunks[k], advance=(k == n - 1), grad_mult=f)
            if self._fused_update and torch.cuda.memory_allocated(self.dev) != before:
                raise RuntimeError('the captured parameter update of exchange chunk %d allocated memory' % k)
            pool_b = gk.pool()
            gbs.append(gk)
        self._graphs = (g1, gbs, g2)
        self._stage1_ready = False

    def _point_grads(self, k):
        """.grad of the parameters of chunk k = views into its bucket (what GradientExchange.wait leaves), without touching the exchange's state."""
        ex = self._ex
        off = 0
        for p in ex.groups[k]:
            p.grad = ex.buckets[k][off:off + p.numel()].view_as(p)
            off += p.numel()
        return ex.grad_mult

    def _agree_on_graphs(self, ok, world_size):
        """Every rank must run the same schedule: if the capture failed anywhere, every rank raises (a rank that silently fell back to eager
        launches would still be correct, but the run would no longer be what its bench line says)."""
        if world_size > 1 and _dist_ready():
            import torch.distributed as dist
            t = torch.tensor([1.0 if ok else 0.0], device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)
        return ok

    def step(self, world_size=1):
        """One iteration; returns the loss tensor.  With several ranks (or force_exchange) and pipeline=True the call RETURNS WITH THE
        TEXTURE CHUNK'S ADAM UPDATE STILL PENDING: kd / ks / normal lag the probe and the vertices by one update until the next step()
        or finish().  Read or change the trained state between steps only through parameters() / set_lr_scale() (they finish() first),
        and call finish() after the last step -- a dropped object with a pending update warns."""
        multi = world_size > 1 or self.force_exchange
        if self.use_graph and not self.force_eager and self._graphs is None and self._eager_steps >= 3:
            err = None
            if multi:
                self._exchange(world_size)
                self._finish_pending()
            try:
                self._capture(world_size)
            except Exception as e:
                err = e
            if multi:
                if not self._agree_on_graphs(err is None, world_size):
                    raise RuntimeError('HIP-graph capture of the iteration failed on at least one rank (this rank: %s); the ranks must '
                                       'all run captured or all eager -- rerun with use_graph=False' % (repr(err) if err else 'ok'))
            elif err is not None:      # one rank: a runtime that cannot capture this iteration keeps running it eagerly (same results)
                import warnings
                warnings.warn('HIP-graph capture of the iteration failed (%s: %s); continuing eagerly' % (type(err).__name__, err))
                self.use_graph, self._graphs = False, None
                torch.cuda.synchronize()
                self.opt.zero_grad(set_to_none=True)
                self._stage1_ready = False
        if multi:
            loss = self._step_multi(world_size)
            self._iter += 1
            return loss
        if self._graphs is not None and not self.force_eager:
            ga = self._graphs[0]
            if isinstance(ga, tuple):       # (rebuild_every: the iteration captured twice, with a rebuild and with a refit)
                k = 1 if self._refit_now() else 0
                ga[k].replay()
                self._iter += 1
                return self._loss_static[k]
            ga.replay()
            self._iter += 1
            return self._loss_static
        self._eager_steps += 1
        loss = self.forward_backward()
        self.allreduce_bytes = 0
        self._update()
        self._tex_grad_guard()
        self._iter += 1
        return loss


def _dist_ready():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()
