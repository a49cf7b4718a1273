#!/usr/bin/env python3
"""bench.py -- MC shadow rays/s and fwd+bwd iterations/s of the direct-lighting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config bob512|spot512x256|dmtet800|hotdog512x256|dmtet64_800|dmtet64_512x256|dmtet64_init512] [--scaling strong|weak]

`--gpus N` with N > 1 spawns N ranks itself (one process per GPU, RCCL = torch.distributed backend "nccl") unless the
process was already started by a launcher (WORLD_SIZE set), e.g.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`):
  bob512      (default; configs[1], the configuration `metric` is quoted on)  bob, 10 688 triangles, 512x512,
              n_samples_x = 8 (64 spp, 128 shadow rays per covered pixel and pass), batch of 8 views per iteration
              (configs/bob.json:8), 256x256 synthetic "sky + suns" probe.
  spot512x256 (configs[2])  spot, 5 856 triangles, metal, 512x512, n_samples_x = 16 (256 spp), batch of 4 views.
  dmtet800    (configs[3] stand-in)  bob subdivided three times = 684 032 triangles (the size of a 128^3 DMTet
              extraction; 43 MB of nodes + triangles: does NOT fit the 4 MB L2s), 800x800, n_samples_x = 8, batch 8.
  hotdog512x256 (configs[4] stand-in)  bob subdivided twice = 171 008 triangles, 512x512, n_samples_x = 16 (256 spp), batch 8
              (configs/nerfactor_hotdog.json:7-8); v_pos is trained like the reference's second (DLMesh) pass -- the COST of the geometry
              gradient, not a converging geometry optimisation (no silhouette term, no regulariser: BENCH_LR_POS).
One step = one optimisation iteration of nvdiffrecmc_amd/trainer.py: update_pdf + BVH rebuild + shading normal +
env-shade fwd + 2x bilateral denoiser + composite + log-sRGB L1 image loss + full backward (the env-shade backward
RE-TRACES every ray, as the reference does) + gradient all-reduce (N > 1) + Adam.  Inputs are resident in HBM before
the timed region.  `value` = shadow rays actually traversed per second, whole job.

Scaling: `strong` (default, north_star's split): ONE batch of views is dealt over the GPUs (8 views on one GPU ... one
view per GPU on eight), one flat RCCL all-reduce of the shared-parameter gradients per iteration.  `weak`: every GPU
renders a whole batch (global batch = batch x N).

Timing protocol: max(20 - W, 0) untimed settle iterations (SURVEY 8d asks for >= 20 warm iterations), then W warm-up
steps, then EXACTLY K steps between barriers (`value`, `ms_per_step`), then an extended phase (>= 50 further steps
and >= 3 s) whose per-step HIP-event times give `median_ms_per_step`.

Extra objects on the JSON line:
  large_mesh   -- (default preset, N = 1) the same iteration on the `dmtet800` preset for a few steps: ms/step, the traversal kernel's
                  rays/s, and its counter-measured HBM traffic / L2 hit rate -- the workload whose tree does not fit the L2s.
  roofline     -- the dominant kernel, env_trace_kernel<false> (persistent-wavefront shadow-ray traversal).  It is
                  VALU-issue bound on cache-resident data (counters below), so `bound` is "valu": `achieved` =
                  active-lane VALU operations per second (SQ_INSTS_VALU x active-lane fraction, rocprofv3 PMC pass of
                  THIS run, / the kernel's average duration from HIP events recorded on the launch stream inside the
                  timed steps), `peak` = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz.  The HBM and L2 fractions of the same
                  kernel are reported beside it (`hbm`: PMC traffic 2*FETCH_SIZE + WRITE_SIZE; `l2`: TCC_REQ x 64 B),
                  and `algorithmic` holds SURVEY 8d's byte model (32-B BVH2 node visits + 36-B triangle tests of the
                  CANONICAL binary walk over the same rays, measured by a counting kernel) over the same duration.
  cpu_baseline -- the reference's own raygen program compiled for the CPU (oracle/_ref; our plain-C restatement when it
                  did not travel), OpenMP over pixels, brute-force visibility, on a pixel subset of the same view.
  config.ms_per_step_cached_visibility / iters_per_sec_cached_visibility -- the same iteration with the forward's visibility bits replayed in
                  backward (exact when forward and backward share the seed; the harness default); never part of `value`.
  config.adam     -- the parameter update alone, tile-sparse (as run) and dense (what mip-mapped textures would cost).
  config.exchange -- (N > 1, or --exchange-world1) what the gradient exchange sent last round, the policy that decided it, the time the main stream
                  stood still for it and the main-stream time of the geometry stage the texture chunk is reduced under.
  config.one_view -- (default preset, N = 1) the per-GPU share of the 8-GPU run: one view in HIP graphs under the several-rank schedule with a
                  one-rank RCCL group on the bytes eight ranks would send, + a priced (not measured) projection to 8 GPUs.
"""
import argparse
import json
import math
import os
import shutil
import socket
import sqlite3
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_parts import (PRESETS, DOMINANT, GEOMETRY_NOTE, TEXTURE_NOTE, BENCH_LR_POS, HBM_PEAK_GBS, VALU_PEAK_TLANEOPS, make_step, algorithmic_bytes,
                               cpu_baseline, cpu_baseline_torch, collect_pmc, find_kernel, valu_figures, mem_figures, other_config_object, adam_object,
                               init_world1, exchange_object, flat_scaling_keys, bvh_policy_note, collect_extras, extras_child, _stdout_to_stderr, _free_port)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', choices=sorted(PRESETS), default='bob512')
    ap.add_argument('--res', type=int, default=None, help='override the preset')
    ap.add_argument('--n-samples-x', type=int, default=None)
    ap.add_argument('--mesh', default=None)
    ap.add_argument('--subdiv', type=int, default=None)
    ap.add_argument('--batch', type=int, default=None, help='views per iteration: in total for strong, per GPU for weak scaling')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong')
    ap.add_argument('--lock-pos', choices=('config', 'on', 'off'), default='config',
                    help='geometry: config = the preset (bob / spot locked as their configs say, the DMTet stand-ins train v_pos), on = locked, off = trained')
    ap.add_argument('--material-set', choices=('full', 'r3'), default='full',
                    help="full = kd + ks + normal textures at the config's texture_res + probe (+ v_pos): the reference's set; r3 = round 3's reduced set (A/B)")
    ap.add_argument('--tex-res', type=int, default=None, help='override the texture resolution of the preset')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes (roofline.frac becomes null)')
    ap.add_argument('--pmc-timeout', type=int, default=240)
    ap.add_argument('--pmc-keep', default=None, help='directory to write the per-kernel counter table of the PMC passes to')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-large-mesh', action='store_true', help='skip the `large_mesh` object (dmtet800: 684 k triangles) of the default N = 1 line')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the `other_configs` objects (spot512x256, hotdog512x256) of the default N = 1 line')
    ap.add_argument('--no-extended', action='store_true', help='skip the extended median phase and the cached-visibility loop')
    ap.add_argument('--exchange', choices=('auto', 'dense', 'sparse'), default='dense',
                    help='gradient exchange of the texture chunk at N > 1: dense = the whole bucket, sparse / auto = the tiles some rank touched (falls back to dense by itself)')
    ap.add_argument('--no-pipeline', action='store_true', help='N > 1: wait for the texture chunk inside the iteration instead of under the next geometry stage')
    ap.add_argument('--exchange-world1', action='store_true',
                    help='N = 1: run the several-rank schedule with a one-rank RCCL group (fixed cost of the exchange path; tile flags OR-ed with the tiles all views of the batch touch)')
    ap.add_argument('--no-one-view', action='store_true', help='skip the `one_view` object (per-GPU share of the 8-GPU run) of the default N = 1 line')
    ap.add_argument('--rebuild-every', type=int, default=None,
                    help='trained geometry: rebuild the BVH every K-th iteration and refit in between (default 8; 1 = a rebuild every iteration, what the reference does)')
    ap.add_argument('--extras-timeout', type=int, default=420, help='seconds the extra objects of the default N = 1 line may take together (they run in a child process)')
    ap.add_argument('--extras-child', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--no-validation', action='store_true', help='skip the `validation_n32` object (n_samples_x = 32, one view) of the default N = 1 line')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), default='auto',
                    help='capture the iteration in HIP graphs (auto: when a rank renders <= 2 views -- the launch-bound regime -- or there are several ranks)')
    return ap.parse_args()


def _spawned(local_rank, world, port, args):
    os.environ.update({'RANK': str(local_rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    run(args)


def main():
    args = parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU fallback')
    if args.extras_child is not None:
        extras_child(args)
        return
    if 'WORLD_SIZE' in os.environ:
        if int(os.environ['WORLD_SIZE']) != args.gpus:
            raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks' % (args.gpus, os.environ['WORLD_SIZE']))
        run(args)
    elif args.gpus > 1:
        # no launcher: start one process per GPU ourselves
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and not os.environ.get('NVDR_BENCH_OVERSUBSCRIBE'):
            raise SystemExit('bench.py: --gpus %d requested but only %d GPU(s) are visible' % (args.gpus, n_dev))
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(args.gpus, _free_port(), args), nprocs=args.gpus, join=True)
    else:
        run(args)


def run(args):
    import torch
    preset = dict(PRESETS[args.config])
    for k, a in (('mesh', args.mesh), ('res', args.res), ('n', args.n_samples_x), ('batch', args.batch), ('subdiv', args.subdiv)):
        if a is not None:
            preset[k] = a
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    n_dev = torch.cuda.device_count()
    if world > n_dev and not os.environ.get('NVDR_BENCH_OVERSUBSCRIBE'):
        raise SystemExit('bench.py: %d ranks but only %d GPU(s) visible (one process per GPU)' % (world, n_dev))
    dev_index = local_rank % n_dev                    # the modulo only matters for oversubscribed 1-GPU dry runs
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('NVDR_BENCH_BACKEND', 'nccl')   # "nccl" is RCCL on ROCm; gloo only for dry runs
        with _stdout_to_stderr():
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
            else:
                dist.init_process_group(backend)
        world = dist.get_world_size()                 # what the collective library actually initialised
    dev = torch.device('cuda', dev_index)

    from nvdiffrecmc_amd.trainer import DirectLightingStep
    from nvdiffrecmc_amd import optixutils as ou
    from nvdiffrecmc_amd.parallel import shard_views
    batch, H, n = preset['batch'], preset['res'], preset['n']
    W = H
    # strong: ONE batch of `batch` views dealt over the GPUs (north_star: "batch=8 views sharded across 8xMI355X");
    # weak: every GPU renders a whole batch.  Every rank seeds its pixels as its slice of the global batch launch.
    n_views = batch * world if args.scaling == 'weak' else batch
    if n_views < world:
        raise SystemExit('bench.py: a batch of %d views cannot be dealt over %d GPUs (--batch / --scaling weak)' % (n_views, world))
    my_views = shard_views(n_views, rank, world)
    if not my_views:
        raise SystemExit('bench.py: rank %d of %d has no view of the batch of %d' % (rank, world, n_views))
    # auto: graphs when a rank renders <= 2 views (launch-bound) and whenever there are several ranks (-1.6 % at 8 views per rank); the
    # one-GPU default stays eager so that the per-stage HIP events behind `roofline` are recorded inside the timed steps themselves
    use_graph = args.graph == 'on' or (args.graph == 'auto' and (len(my_views) <= 2 or world > 1))
    if args.pmc_child:
        use_graph = False
    lock_pos = preset.get('lock_pos', True) if args.lock_pos == 'config' else (args.lock_pos == 'on')
    if args.material_set == 'r3':
        lock_pos = True
    forced = bool(args.exchange_world1 and world == 1 and not args.pmc_child)
    if forced:
        init_world1(dev)
    step = make_step(preset, args, dev, my_views, n_views, lock_pos, pixel_index_offset=my_views[0] * H * W, retrace_backward=True, use_graph=use_graph,
                     force_exchange=forced, union_views=(list(range(n_views)) if forced else None))

    if args.pmc_child:          # under rocprofv3: a few plain iterations, nothing else
        for _ in range(args.warmup + args.steps):
            step.step(world)
        torch.cuda.synchronize()
        return

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    settle = max(0, 20 - args.warmup)                 # >= 20 warm iterations before anything is timed
    if use_graph and settle + args.warmup < 4:
        settle = 4 - args.warmup                      # the graphs are captured after three eager iterations
    for _ in range(settle + args.warmup):
        step.step(world)
    # per-stage HIP-event timing recorded by the library on the launch stream itself (ring of the last launches); a
    # replayed graph cannot carry them, so in graph mode the stage times come from an eager phase after the timed ones
    if not use_graph:
        step.ctx.set_profiling(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    multi = world > 1 or forced
    if multi:
        step.measure_exposed = True        # a pair of HIP events around every wait on the exchange (a microsecond each)
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step.step(world)
        b.record()
    step.finish()               # (several ranks: the pipelined texture update of the last step belongs to the timed region)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    step_ms = [a.elapsed_time(b) for a, b in ev]
    graph_active = step._graphs is not None            # False also when the capture failed and the harness fell back to eager
    if not use_graph:
        n_f, (gen_ms, trace_ms, shade_ms) = step.ctx.stage_times(backward=False)
        n_b, (bgen_ms, btrace_ms, bshade_ms) = step.ctx.stage_times(backward=True)
        step.ctx.set_profiling(False)

    # extended phase: more samples of the same iteration for the median (>= 50 steps and >= 3 s), then the same iteration
    # with the forward's visibility bits replayed in backward (identical gradients, no second traversal; an extra, never `value`)
    ext_ms, dt2, k2 = [], None, 0
    if not args.no_extended:
        t_ext = time.perf_counter()
        while True:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            step.step(world)
            b.record()
            b.synchronize()
            ext_ms.append(a.elapsed_time(b))
            more = len(ext_ms) < 50 or (time.perf_counter() - t_ext < 3.0 and len(ext_ms) < 2000)
            if dist is not None:     # all ranks must run the same number of collectives: agree on when to stop
                flag = torch.tensor([1.0 if more else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = flag.item() != 0.0
            if not more:
                break
    exposed = step.exposed_ms() if multi else None
    step.measure_exposed = False
    if use_graph:                                     # leave graph mode for good: stage times of the same kernels, eagerly
        step.finish()
        step.force_eager = True
        for _ in range(2):
            step.step(world)
        step.ctx.set_profiling(True)
        for _ in range(10):
            step.step(world)
        barrier()
        n_f, (gen_ms, trace_ms, shade_ms) = step.ctx.stage_times(backward=False)
        n_b, (bgen_ms, btrace_ms, bshade_ms) = step.ctx.stage_times(backward=True)
        step.ctx.set_profiling(False)
    if not args.no_extended:
        step.retrace_backward = False
        k2 = max(5, args.steps // 2)
        for _ in range(2):
            step.step(world)
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            step.step(world)
        barrier()
        dt2 = max_over_ranks(time.perf_counter() - t1)
        step.retrace_backward = True

    S = n * n
    # counting build of the same forward kernel on this rank's views -> rays actually traversed (dead samples --
    # dot(n, wi) <= 0, zero through the BSDF's own gates -- are never traced), wide-walk box tests, and the canonical
    # binary walk's node visits / triangle tests over the same rays (the algorithmic-byte model)
    light = step.light
    with torch.no_grad():
        from nvdiffrecmc_amd import renderutils as ru
        m = step.mask[..., None]
        _, ro, _, nrm, _, kd, ks = step.shade_inputs()        # the tensors optix_env_shade gets inside the iteration
        P, n_box, n_tri, n_traced = ou.ops.env_shade_traversal_counts(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks,
                                                                     light.base, light._pdf, light.rows[:, 0], light.cols,
                                                                     n_samples_x=n, rnd_seed=0)
        bvh2_nodes, bvh2_tris, bvh2_rays = ou.ops.env_shade_traversal_counts.bvh2
        clock_mhz = ou.ops.env_shade_traversal_counts.clock_mhz
    rays_pass = torch.tensor([step.rays_per_pass(), n_traced], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(rays_pass, op=dist.ReduceOp.SUM)
    queries_step_total = 2.0 * float(rays_pass[0].item())  # forward + backward shadow-ray queries (2*S per covered pixel), all ranks
    rays_step_total = 2.0 * float(rays_pass[1].item())     # of which traversed: forward + re-traced backward, all ranks
    fwd_ms = gen_ms + trace_ms + shade_ms
    n_tris = int(step.mesh['t_pos_idx'].shape[0])

    if rank == 0:
        probe = light.base.shape[0]
        bytes_fwd, bytes_trace, b_trav = algorithmic_bytes(step.nv, H, W, P, S, probe, bvh2_nodes, bvh2_tris, n_traced)
        R = 2 * S * P
        roof = {'bound': 'valu', 'achieved': None, 'peak': VALU_PEAK_TLANEOPS, 'unit': 'T lane-ops/s', 'frac': None, 'traffic': None,
                'kernel': DOMINANT, 'kernel_ms_hip_events': trace_ms, 'launches_timed': n_f,
                'rays_per_launch': n_traced, 'kernel_rays_per_sec': n_traced / (trace_ms * 1e-3),
                'why_valu': 'the active-lane VALU rate is the ceiling with a datasheet number this kernel is nearest to (its HBM and L2 fractions -- hbm, l2 below -- are '
                            'small: divergent traversal of a cache-resident tree); peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz.  It is NOT a claim that the issue port is '
                            'full: the phase-clock builds of round 6 (profiles/r06_trace_phase_cycles.md) put half of a wavefront\'s cycles into the node FETCH (four '
                            '16-byte requests per lane and step, waited for) and under a tenth into the box arithmetic, and doubling those requests costs +22 % here, '
                            '+40 % on 684 k triangles (profiles/r06_ab_trace_split_coop_dup.md; DESIGN.md 4.6, 5)',
                'shader_clock_mhz_counting_launch': clock_mhz,
                'algorithmic': {'model': 'SURVEY 8d: 32 B per BVH2 node visit + 36 B per triangle test of the canonical binary any-hit walk '
                                         '(counting kernel over the same live rays; equals a CPU walk of the exported tree, tests/test_gpu_bvh.py) '
                                         '+ 21 B per traversed ray + 16 B per pixel of ray stream',
                                'bytes_per_launch': bytes_trace, 'traversal_bytes_per_launch': b_trav,
                                'bvh2_node_visits_per_ray': bvh2_nodes / max(bvh2_rays, 1), 'bvh2_tri_tests_per_ray': bvh2_tris / max(bvh2_rays, 1),
                                'wide_walk_box_tests_per_ray': n_box / max(n_traced, 1), 'wide_walk_tri_tests_per_ray': n_tri / max(n_traced, 1),
                                'oct_walk_node_steps_per_ray': ou.ops.env_shade_traversal_counts.node_steps / max(n_traced, 1),
                                'triangle_test_batch_fill': (ou.ops.env_shade_traversal_counts.leaf_batches[1]
                                                             / max(64 * ou.ops.env_shade_traversal_counts.leaf_batches[0], 1)),
                                'GBs': bytes_trace / (trace_ms * 1e-3) / 1e9,
                                'frac_of_hbm_peak_if_it_were_hbm_traffic': bytes_trace / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                'forward_pass': {'gen_ms': gen_ms, 'trace_ms': trace_ms, 'shade_ms': shade_ms, 'algorithmic_bytes': bytes_fwd,
                                 'rays_per_sec': n_traced / (fwd_ms * 1e-3)},
                'backward_pass': {'gen_ms': bgen_ms, 'trace_ms': btrace_ms, 'shade_and_light_gradient_ms': bshade_ms, 'launches_timed': n_b}}
        if world == 1 and not args.no_pmc:
            t_p = time.perf_counter()
            counters, note = collect_pmc(args, keep_dir=args.pmc_keep)
            roof['pmc_seconds'] = time.perf_counter() - t_p
            if note:
                roof['pmc_note'] = note
            c = find_kernel(counters, DOMINANT) if counters else None
            if c:
                # The child ran 3 iterations + the target render = 7 env-shade passes.  A launch whose ray stream is cut into chunks
                # (larger than the context's byte budget, 8 GiB by default) dispatches the kernel once per non-empty chunk and pass;
                # the counter sums are per DISPATCH, the HIP-event time is per pass: bring the counters to the pass.
                chunks = max(1, int(round(c.get('dispatches_pass0', 7) / 7.0)))
                if chunks > 1:
                    c = {k: (v_ * chunks if not k.startswith('dispatches_pass') else v_) for k, v_ in c.items()}
                    roof['dispatches_per_launch'] = chunks
                v = valu_figures(c, trace_ms)
                mem = mem_figures(c, trace_ms)
                if v:
                    roof['achieved'] = v['achieved_Tlaneops']
                    roof['frac'] = v['frac_of_lane_peak']
                    roof['valu'] = v
                roof['traffic'] = mem.get('hbm_bytes')
                roof['hbm'] = {k: mem[k] for k in ('hbm_bytes', 'fetch_bytes_corrected', 'write_bytes', 'hbm_GBs', 'hbm_frac') if k in mem}
                roof['l2'] = {k: mem[k] for k in ('l2_requests', 'l2_hit', 'l2_GBs_at_64B_per_request', 'l2_frac') if k in mem}
                roof['traffic_source'] = 'rocprofv3 --kernel-trace --pmc passes of this run (bench.py --pmc-child, %d dispatches)' % int(c.get('dispatches_pass1', 0))
                # the other env-shade kernels from the same passes (durations: HIP-event stage times, backward stage 3 includes the gather)
                others = {}
                # (S = 64 launches run the shading kernels that queue the light samples across pixels, env_shade_queue_kernel; S > 64 backward env_shade_local_kernel)
                for label, needle, ms in (('env_shade_kernel<backward>', ('env_shade_queue_kernel<true', 'env_shade_local_kernel<true', 'env_shade_kernel<true'), None), ('env_gen_kernel', 'env_gen_kernel', gen_ms),
                                          ('env_shade_kernel<forward>', ('env_shade_queue_kernel<false', 'env_shade_local_kernel<false', 'env_shade_kernel<false'), shade_ms),
                                          ('light_grad_block_kernel', 'light_grad_block_kernel', None)):
                    oc = find_kernel(counters, needle)
                    kname = next((nm for nm, cc in counters.items() if cc is oc), None)
                    if oc and chunks > 1:
                        oc = {k: (v_ * chunks if not k.startswith('dispatches_pass') else v_) for k, v_ in oc.items()}
                    if oc:
                        mm = mem_figures(oc, ms)
                        vv = valu_figures(oc, ms) if ms else None
                        others[label] = {'kernel': kname, 'hbm_bytes': mm.get('hbm_bytes'), 'hbm_GBs': mm.get('hbm_GBs'), 'l2_hit': mm.get('l2_hit'),
                                          'valu_wave_instructions': oc.get('SQ_INSTS_VALU'),
                                          'active_lane_fraction': (oc['SQ_THREAD_CYCLES_VALU'] / (64.0 * oc['SQ_ACTIVE_INST_VALU'])
                                                                   if oc.get('SQ_ACTIVE_INST_VALU') else None),
                                          'frac_of_lane_peak': vv['frac_of_lane_peak'] if vv else None}
                roof['other_kernels'] = others
        else:
            roof['pmc_note'] = 'counter passes skipped (%s)' % ('--no-pmc' if args.no_pmc else 'N > 1: counters are collected at N = 1 only')
        med = statistics.median(ext_ms) if ext_ms else statistics.median(step_ms)
        out = {
            'metric': preset['metric'],
            'value': rays_step_total * args.steps / dt,       # TRAVERSED rays only
            'unit': 'rays/s',
            'shadow_ray_queries_per_sec': queries_step_total * args.steps / dt,   # 2*S per covered pixel per pass, traversed or not
            'iters_per_sec': args.steps / dt,
            'iters_per_sec_cached_visibility': (k2 / dt2) if dt2 else None,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'settle_steps_before_warmup': settle,
            'hip_graph': bool(graph_active),
            'ms_per_step': dt / args.steps * 1e3,
            'median_ms_per_step': med, 'median_over_steps': len(ext_ms) if ext_ms else len(step_ms),
            'min_ms_per_step': min(ext_ms or step_ms), 'max_ms_per_step': max(ext_ms or step_ms),
            # steps that took more than twice the median (straggler flag: until the end of round 2 an exactly axis-parallel ray
            # could hold a launch for 0.3-0.4 s, profiles/r02_slow_mode.md); `value` is the mean over the timed steps and includes them
            'steps_over_twice_the_median': sum(1 for v in (ext_ms or step_ms) if v > 2.0 * med),
            'timed_steps_over_twice_the_median': sum(1 for v in step_ms if v > 2.0 * med),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, batch of %d views per iteration (%s), LBVH + HIP traversal + GGX shading + bilateral denoiser + log-sRGB L1 loss, fwd+bwd+Adam'
                                   % (preset['what'], n_views, 'one batch dealt over the GPUs' if args.scaling == 'strong' else '%d per GPU' % batch),
                       'preset': args.config, 'mesh_triangles': n_tris, 'covered_pixels_rank0': P,
                       'shadow_ray_queries_per_pass_rank0': R, 'rays_traversed_per_pass_rank0': n_traced,
                       'dead_samples': '%.1f%% of the queries have dot(n,wi)<=0, are zero through the BSDF gates whatever their visibility and are answered without traversal (outputs bit-identical; NVDR_DEBUG=8 traces them); value counts traversed rays only' % (100.0 * (1.0 - n_traced / max(R, 1))),
                       'views_per_iteration': n_views, 'views_rank0': step.nv, 'probe': '%dx%d E1' % (probe, probe),
                       'backward': 're-traces all shadow rays (what the reference\'s backward does; `value` / `ms_per_step` are defined on this iteration)',
                       # the same iteration with the forward's visibility bits replayed in backward -- exact whenever forward and backward share the seed
                       # (train.py:547 decorrelated=False, render.py:112-116; per-pixel gradients bit-identical, tests/test_gpu_env_shade.py) and the
                       # default of trainer.DirectLightingStep; measured after the timed steps, never part of `value`
                       'ms_per_step_cached_visibility': (dt2 / k2 * 1e3) if dt2 else None,
                       'iters_per_sec_cached_visibility': (k2 / dt2) if dt2 else None,
                       'texture_filter': TEXTURE_NOTE,
                       'parallelism': 'dp%d (%d views per GPU)' % (world, step.nv),
                       'trained_parameters': {nm: list(p.shape) for nm, p in zip(step.param_names, step.params)},
                       'parameter_bytes': int(sum(p.numel() for p in step.params) * 4),
                       'geometry': ('locked (lock_pos): G-buffer of the fixed views rendered once, BVH rebuilt every iteration as the reference does' if lock_pos else
                                    'trained (v_pos, lr %g): BVH, vertex normals / tangents and the G-buffer follow the moving vertices every iteration; ' % BENCH_LR_POS + GEOMETRY_NOTE),
                       'bvh': bvh_policy_note(args, lock_pos),
                       'allreduce_bytes_per_step': getattr(step, 'allreduce_bytes', 0)},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline and not preset['subdiv'] and n_tris <= 20000:
            # bounded sample: ~4e10 ray-triangle tests (10-30 s on the box's host cores) = covered pixels x 4S rays x triangles
            stride = max(2, int(math.ceil(math.sqrt(0.23 * H * W * 4 * S * n_tris / 4e10))))
            out['cpu_baseline'] = cpu_baseline(preset['mesh'], H, n, 0, n_views, stride=stride)
        else:
            out['cpu_baseline'] = None   # N > 1, --no-cpu-baseline, or a large mesh (brute force over 140-684 k triangles is not a bounded sample)
        if out['cpu_baseline'] is not None:
            out['cpu_baseline']['note'] = ('this is the reference\'s own CUDA raygen program compiled for the host and run under OpenMP on every core '
                                           '(kind "reference"), NOT the brute-force PyTorch-CPU path BASELINE.json sketches: the same arithmetic, a '
                                           'faster CPU implementation of it than torch ops would be')
        if world == 1 and not args.no_cpu_baseline and args.config == 'bob512':
            try:
                out['cpu_baseline_torch'] = cpu_baseline_torch()
            except Exception as e:
                out['cpu_baseline_torch'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if multi:
            out['config']['exchange'] = exchange_object(step, exposed)
        if not args.pmc_child and step._graphs is None and world == 1 and not forced:
            try:
                out['config']['adam'] = adam_object(step)
            except Exception as e:
                out['config']['adam'] = {'error': '%s: %s' % (type(e).__name__, e)}
        default_line = (world == 1 and args.config == 'bob512' and not args.pmc_child and args.res is None and args.subdiv is None and args.batch is None
                        and args.mesh is None and args.n_samples_x is None and not forced)
        eight_ms = dt / args.steps * 1e3

        if default_line:
            # The extra objects run in a CHILD process behind a deadline, one JSON line per object as it completes: an object that raises, hangs
            # or takes the GPU down costs that object (and the ones behind it), never the headline line above.  (Round 6: one of them hung a GPU
            # queue during development; a bench line that never arrives would be the worst outcome of all.)
            step = None
            torch.cuda.empty_cache()
            got, note = collect_extras(args, eight_ms)
            for key, val in got:
                if key == 'one_view':
                    out['config']['one_view'] = val
                    out['config'].update(flat_scaling_keys('one_view', val))
                elif key.startswith('other_configs/'):
                    out.setdefault('other_configs', {})[key.split('/', 1)[1]] = val
                elif key.startswith('large_mesh_regular/one_view_'):
                    tag = key.rsplit('_', 1)[1]
                    (out.get('large_mesh_regular') or out.setdefault('large_mesh_regular', {}))['one_view_' + tag] = val
                    out['config'].update(flat_scaling_keys('large_mesh_one_view_' + tag, val))
                else:
                    out[key] = val
            if note:
                out['extras_note'] = note
        if forced:      # (the one-rank group of --exchange-world1: torn down BEFORE the line is written -- RCCL's exit-time teardown has cut a line short)
            step = None
            import torch.distributed as _d
            with _stdout_to_stderr():
                if _d.is_initialized():
                    _d.destroy_process_group()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        with _stdout_to_stderr():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
