#!/usr/bin/env python3
"""bench.py -- MC shadow rays/s and fwd+bwd iterations/s of the direct-lighting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): bob mesh (10 688 triangles), 512x512, n_samples_x = 8 (64 spp,
128 shadow rays per covered pixel per pass), 256x256 synthetic "sky + suns" probe, one camera view
per GPU (weak scaling).  One step = one optimisation iteration of nvdiffrecmc_amd/trainer.py:
update_pdf + BVH rebuild + shading normal + env-shade fwd + 2x bilateral denoiser + combine +
log-sRGB L1 image loss + full backward (the env-shade backward RE-TRACES every ray, as the
reference does) + gradient all-reduce (N > 1) + Adam.  Inputs are resident in HBM before the timed
region.  `value` = shadow rays actually traced per second, whole job.

Extra objects on the JSON line:
  roofline     -- the dominant kernel, env_trace_kernel (persistent-wavefront shadow-ray traversal): algorithmic
                  bytes per launch (32 B per box test + 36 B per triangle test, counts measured by the
                  counting build of the same kernel, + its 17 B/ray stream) over its average duration from HIP
                  events the library records on the launch stream inside the timed steps, vs 8 TB/s.
  cpu_baseline -- the CPU oracle (plain C, OpenMP over pixels, brute-force visibility) on a 1/4 pixel
                  subset of the same view, fwd + bwd, on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling


def algorithmic_bytes(N, H, W, P, S, probe, n_box, n_tri, n_traced=None):
    """B = B_stream + B_tables + B_trav for one forward pass, and the share the traversal kernel
    (env_trace_kernel) moves: B_trav + its ray stream (16 B direction+pdf in, 1 B visibility out per ray, 16 B origin per pixel)."""
    NHW = N * H * W
    R = 2 * S * P
    b_stream = 4 * NHW + 60 * P + 24 * NHW
    m = (probe - 1).bit_length() + 1  # ceil(log2(size-1)) + 1 bisection steps
    b_tables = P * S * (8 + 4 * (m + 2) + 4 * (m + 2) + 2 * (4 + 12))
    b_trav = 32 * n_box + 36 * n_tri
    # every slot of the stream is read (16 B: direction + pdf sum, the sign bit flags a dead sample); only traversed
    # rays fetch their pixel's origin and write a visibility byte
    b_trace_kernel = b_trav + 16 * R + 1 * (R if n_traced is None else n_traced) + 16 * P
    return b_stream + b_tables + b_trav, b_trace_kernel, b_trav


def cpu_baseline(res, n, view, n_views, stride=2):
    """Oracle fwd+bwd on every stride-th pixel in x and y of the same view; returns the JSON object."""
    from oracle import oracle as orc, scene_cpu
    nt = orc.max_threads()
    inp = scene_cpu.make_inputs('bob', res, res, n, view=view, n_views=n_views, n_threads=nt)
    m = inp['mesh']
    sub = torch.zeros_like(inp['mask'])
    sub[:, ::stride, ::stride] = inp['mask'][:, ::stride, ::stride]
    inp['mask'] = sub
    kw = scene_cpu.shade_kwargs(inp)
    g = torch.Generator().manual_seed(0)
    dg, sg = torch.rand(1, res, res, 3, generator=g), torch.rand(1, res, res, 3, generator=g)
    # the REFERENCE's own raygen program compiled for the CPU (oracle/_ref, prebuilt where /root/reference exists) when
    # it travelled with the repo, otherwise our plain-C restatement of it; visibility is a brute-force loop in both
    impl, kind, what = ('ref', 'reference', 'oracle/_ref: the reference envsampling/kernel.cu built for the CPU') if orc.have_ref() \
        else ('oracle', 'port', 'oracle/nvdr_oracle.c')
    t0 = time.perf_counter()
    f = orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=0, n_threads=nt, impl=impl)
    orc.env_shade(m['v_pos'], m['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n, rnd_seed=0, diff_grad=dg, spec_grad=sg, n_threads=nt, impl=impl)
    dt = time.perf_counter() - t0
    rays = 2 * (2 * n * n * f['covered'])
    return {'value': rays / dt, 'unit': 'rays/s', 'cores': nt, 'kind': kind,
            'sample': '%s, env-shade fwd+bwd (OpenMP over pixels, brute-force visibility over 10688 triangles), every %dth pixel '
                      'in x and y of the %dx%d view (%d covered pixels, %d rays), %.1f s' % (what, stride, res, res, f['covered'], rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--res', type=int, default=512)
    ap.add_argument('--n-samples-x', type=int, default=8)
    ap.add_argument('--mesh', default='bob')
    ap.add_argument('--batch', type=int, default=8, help='views per iteration (configs/bob.json:8): per GPU for weak, in total for strong scaling')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU fallback')
    dev_index = local_rank % torch.cuda.device_count()   # one process per GPU; the modulo only matters for 1-GPU dry runs
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('NVDR_BENCH_BACKEND', 'nccl')   # "nccl" is RCCL on ROCm; gloo only for dry runs
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device('cuda', dev_index)

    from nvdiffrecmc_amd.trainer import DirectLightingStep
    from nvdiffrecmc_amd import optixutils as ou
    # Every GPU renders the batch of the reference config (configs/bob.json:8: 8 views per iteration, stacked along N like
    # the reference's render()): per-GPU work is fixed, the global batch is 8 * world views ("weak").  --scaling strong
    # instead shards ONE batch of 8 views over the GPUs (8 on one GPU ... 1 view per GPU on eight: north_star's
    # "batch=8 views sharded across 8xMI355X").  Every rank seeds its pixels as its slice of the global batch launch.
    from nvdiffrecmc_amd.parallel import shard_views
    n_views = args.batch * world if args.scaling == 'weak' else max(world, args.batch)
    H = W = args.res
    my_views = shard_views(n_views, rank, world)
    if not my_views:
        raise SystemExit('bench.py: rank %d of %d has no view of the batch of %d' % (rank, world, n_views))
    step = DirectLightingStep(args.mesh, args.res, args.n_samples_x, view=my_views, n_views=n_views, device=dev,
                              pixel_index_offset=my_views[0] * H * W, retrace_backward=True)

    # per-stage HIP-event timing recorded by the library on the launch stream itself (ring of the last 128 launches)
    step.ctx.set_profiling(True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.step(world)
    step.ctx.set_profiling(True)   # clears the ring
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.step(world)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_f, (gen_ms, trace_ms, shade_ms) = step.ctx.stage_times(backward=False)
    n_b, (bgen_ms, btrace_ms, bshade_ms) = step.ctx.stage_times(backward=True)
    step.ctx.set_profiling(False)

    # second, shorter timed loop: the same iteration with the forward pass's visibility bits replayed in backward
    # (identical gradients, no second traversal) -- reported as an extra, never as `value`
    step.retrace_backward = False
    k2 = max(5, args.steps // 2)
    for _ in range(2):
        step.step(world)
    barrier()
    t1 = time.perf_counter()
    for _ in range(k2):
        step.step(world)
    barrier()
    dt2 = time.perf_counter() - t1
    if dist is not None:
        tt = torch.tensor([dt2], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt2 = float(tt.item())
    step.retrace_backward = True

    S = args.n_samples_x ** 2
    # counting build of the same forward kernel on this rank's view -> measured traversal work and the number of rays
    # actually traversed (dead samples -- dot(n, wi) <= 0, zero through the BSDF's own gates -- are never traced)
    light = step.light
    with torch.no_grad():
        from nvdiffrecmc_amd import renderutils as ru
        m = step.mask[..., None]
        kd = step.kd_tex[step.texel].view(step.nv, H, W, 3) * m  # same values as the step's kd image
        ks = step.ks.view(1, 1, 1, 3) * m
        nrm = ru.prepare_shading_normal(step.gb_pos, step.view_pos, None, step.gb_smooth_nrm, step.gb_tangent, step.gb_geom_nrm)
        ro = step.gb_pos + nrm * 0.001
        P, n_box, n_tri, n_traced = ou.ops.env_shade_traversal_counts(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks,
                                                                     light.base, light._pdf, light.rows[:, 0], light.cols,
                                                                     n_samples_x=args.n_samples_x, rnd_seed=0)
    rays_pass = torch.tensor([step.rays_per_pass(), n_traced], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(rays_pass, op=dist.ReduceOp.SUM)
    queries_step_total = 2.0 * float(rays_pass[0].item())  # forward + backward shadow-ray queries (2*S per covered pixel), all ranks
    rays_step_total = 2.0 * float(rays_pass[1].item())     # of which traversed: forward + re-traced backward, all ranks
    fwd_ms = gen_ms + trace_ms + shade_ms

    if rank == 0:
        probe = light.base.shape[0]
        bytes_fwd, bytes_trace, b_trav = algorithmic_bytes(step.nv, H, W, P, S, probe, n_box, n_tri, n_traced)
        achieved = bytes_trace / (trace_ms * 1e-3) / 1e9
        # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (counters cannot be read in-process)
        traffic, traffic_src = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')))
            traffic = (2.0 * pm['FETCH_SIZE_KB_per_launch'] + pm['WRITE_SIZE_KB_per_launch']) * 1024.0
            traffic_src = pm['source']
        except Exception:
            pass
        R = 2 * S * P
        out = {
            'metric': 'MC shadow rays/sec (fwd+bwd train iteration, 512x512 64spp bob mesh)',
            'value': rays_step_total * args.steps / dt,       # TRAVERSED rays only
            'unit': 'rays/s',
            'shadow_ray_queries_per_sec': queries_step_total * args.steps / dt,   # 2*S per covered pixel per pass, traversed or not
            'iters_per_sec': args.steps / dt,
            'iters_per_sec_cached_visibility': k2 / dt2,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'bob.json %dx%d, %d spp (n_samples_x=%d), batch of %d views per iteration (configs/bob.json:8; %s), LBVH + HIP traversal + GGX shading + bilateral denoiser + log-sRGB L1 loss, fwd+bwd+Adam'
                                   % (H, W, S, args.n_samples_x, n_views, '%d per GPU' % args.batch if args.scaling == 'weak' else 'one batch sharded over the GPUs'),
                       'mesh_triangles': int(step.mesh['t_pos_idx'].shape[0]), 'covered_pixels_rank0': P,
                       'shadow_ray_queries_per_pass_rank0': R, 'rays_traversed_per_pass_rank0': n_traced,
                       'dead_samples': '%.1f%% of the queries have dot(n,wi)<=0, are zero through the BSDF gates whatever their visibility and are answered without traversal (outputs bit-identical; NVDR_DEBUG=8 traces them); value counts traversed rays only' % (100.0 * (1.0 - n_traced / R)),
                       'views_per_iteration': n_views, 'views_rank0': step.nv, 'probe': '%dx%d E1' % (probe, probe),
                       'backward': 're-traces all shadow rays', 'parallelism': 'dp%d (%d views per GPU)' % (world, step.nv)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel': 'env_trace_kernel<false>',
                         'kernel_ms_hip_events': trace_ms, 'launches_timed': n_f,
                         'algorithmic_bytes_per_launch': bytes_trace, 'traversal_bytes_per_launch': b_trav,
                         'box_tests_per_ray': n_box / n_traced, 'tri_tests_per_ray': n_tri / n_traced,
                         'rays_per_launch': n_traced, 'kernel_rays_per_sec': n_traced / (trace_ms * 1e-3),
                         'forward_pass': {'gen_ms': gen_ms, 'trace_ms': trace_ms, 'shade_ms': shade_ms,
                                          'algorithmic_bytes': bytes_fwd, 'achieved_GBs': bytes_fwd / (fwd_ms * 1e-3) / 1e9,
                                          'rays_per_sec': n_traced / (fwd_ms * 1e-3)},
                         'backward_pass': {'gen_ms': bgen_ms, 'trace_ms': btrace_ms, 'shade_ms': bshade_ms, 'launches_timed': n_b}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.res, args.n_samples_x, 0, n_views)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
