#!/usr/bin/env python3
"""bench.py -- MC shadow rays/s and fwd+bwd iterations/s of the direct-lighting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config bob512|spot512x256|dmtet800|hotdog512x256] [--scaling strong|weak]

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

HBM_PEAK_GBS = 8000.0      # MI355X spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth, same guide
N_CUS, SIMDS_PER_CU, LANES_PER_SIMD, CLOCK_GHZ = 256, 4, 32, 2.4
VALU_PEAK_TLANEOPS = N_CUS * SIMDS_PER_CU * LANES_PER_SIMD * CLOCK_GHZ * 1e9 / 1e12   # 78.6 T lane-ops/s (x2 flop = 157.3 TFLOP/s)

# Position learning rate of the geometry-unlocked presets.  The reference moves v_pos with lr 0.005 (configs/nerf_lego.json:9, second pass) under
# a Laplacian regulariser and silhouette gradients; this harness has neither, and Adam with a Monte-Carlo-noisy gradient random-walks the
# vertices by lr per step: at 0.01 the mesh is noise after 30 iterations and the traversal 8 x slower (252 ms per dmtet800 iteration, measured).
# 1e-5 keeps the mesh a mesh over the few hundred iterations of a bench run (< 0.003 units of drift); the work per iteration does not depend on it.
BENCH_LR_POS = 1e-5

PRESETS = {
    # lock_pos / tex_res: the config's own keys (configs/bob.json:6,14; spot_metal.json; nerf_lego / nerfactor_hotdog train the geometry:
    # their second pass runs DLMesh with v_pos as a parameter, geometry/dlmesh.py:28-38)
    # ks_min: configs/bob.json:10 and spot_metal.json:12 say [0, 0.1, 0]; nerf_lego / nerfactor_hotdog keep the default of train.py:550 (0.08)
    'bob512': dict(mesh='bob', res=512, n=8, batch=8, subdiv=0, lock_pos=True, tex_res=1024, ks_min=(0.0, 0.1, 0.0),
                   metric='MC shadow rays/sec (fwd+bwd train iteration, 512x512 64spp bob mesh)',
                   what='bob.json 512x512, 64 spp (n_samples_x=8)'),
    # spot_metal.json:8,20: texture_res 512, no_perturbed_nrm (no normal map in the trained set)
    'spot512x256': dict(mesh='spot', res=512, n=16, batch=4, subdiv=0, lock_pos=True, tex_res=512, ks_min=(0.0, 0.1, 0.0), perturbed_nrm=False,
                        metric='MC shadow rays/sec (fwd+bwd train iteration, 512x512 256spp spot_metal)',
                        what='spot_metal.json 512x512, 256 spp (n_samples_x=16)'),
    'dmtet800': dict(mesh='bob', res=800, n=8, batch=8, subdiv=3, lock_pos=False, tex_res=1024,
                     metric='MC shadow rays/sec (fwd+bwd train iteration, 800x800 64spp, 684k-triangle DMTet-sized mesh)',
                     what='nerf_lego.json stand-in: bob subdivided 3x (684 032 triangles), 800x800, 64 spp (n_samples_x=8)'),
    'hotdog512x256': dict(mesh='bob', res=512, n=16, batch=8, subdiv=2, lock_pos=False, tex_res=1024,
                          metric='MC shadow rays/sec (fwd+bwd train iteration, 512x512 256spp, 171k-triangle DMTet-sized mesh)',
                          what='nerfactor_hotdog.json stand-in: bob subdivided 2x (171 008 triangles, the size DMTet extracts from a 128^3 '
                               'grid), 512x512, 256 spp (n_samples_x=16)'),
}
DOMINANT = 'env_trace_kernel<false>'
GEOMETRY_NOTE = ('v_pos trained at lr %g: this measures the WORK SHAPE of geometry training (BVH / vertex frames / G-buffer rebuilt from the moving vertices, '
                 'interpolation adjoint, v_pos in the exchange and in Adam), not a converging geometry optimisation -- the reference adds silhouette gradients '
                 '(dr.antialias, render.py:290) and a Laplacian regulariser (geometry/dlmesh.py:57-76), both outside the path' % BENCH_LR_POS)
TEXTURE_NOTE = ('trained textures are sampled at the NEAREST texel (render/texture.py:57-68 uses dr.texture linear-mipmap-linear, outside the path): only the '
                'texels some covered pixel looks up receive gradient, which the tile-sparse Adam and the tile-sparse exchange exploit; with the mip chain of the '
                'reference every texel would receive gradient -- config.adam.dense_ms and exchange mode "dense" are the like-for-like figures')


def make_step(pre, args, dev, views, n_views, lock_pos, **kw):
    """The iteration object of one preset (trainer.DirectLightingStep) with the config's own keys."""
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    return DirectLightingStep(pre['mesh'], pre['res'], pre['n'], view=views, n_views=n_views, device=dev, subdiv=pre['subdiv'],
                              material_set=args.material_set, tex_res=args.tex_res or pre.get('tex_res', 1024), optimize_geometry=not lock_pos,
                              lr_pos=BENCH_LR_POS, ks_min=pre.get('ks_min', (0.0, 0.08, 0.0)), perturbed_nrm=pre.get('perturbed_nrm', True),
                              **dict(dict(exchange_mode=args.exchange, pipeline=not args.no_pipeline), **kw))


def algorithmic_bytes(N, H, W, P, S, probe, bvh2_nodes, bvh2_tris, n_traced):
    """B = B_stream + B_tables + B_trav for one forward pass, and the share the traversal kernel moves:
    B_trav (32 B per BVH2 node visit + 36 B per triangle test of the canonical binary walk) + its ray stream (16 B
    direction + pdf per slot in, 4 B list entry + 1 B visibility per traversed ray, 16 B origin per pixel)."""
    NHW = N * H * W
    R = 2 * S * P
    b_stream = 4 * NHW + 60 * P + 24 * NHW
    m = (probe - 1).bit_length() + 1  # ceil(log2(size-1)) + 1 bisection steps
    b_tables = P * S * (8 + 4 * (m + 2) + 4 * (m + 2) + 2 * (4 + 12))
    b_trav = 32 * bvh2_nodes + 36 * bvh2_tris
    b_trace_kernel = b_trav + (16 + 4 + 1) * n_traced + 16 * P
    return b_stream + b_tables + b_trav, b_trace_kernel, b_trav


def cpu_baseline(mesh_name, res, n, view, n_views, stride=2):
    """Oracle fwd+bwd on every stride-th pixel in x and y of the same view; returns the JSON object."""
    import torch
    from oracle import oracle as orc, scene_cpu
    nt = orc.max_threads()
    inp = scene_cpu.make_inputs(mesh_name, res, res, n, view=view, n_views=n_views, n_threads=nt)
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
            'sample': '%s, env-shade fwd+bwd (OpenMP over pixels, brute-force visibility over %d triangles), every %dth pixel '
                      'in x and y of the %dx%d view (%d covered pixels, %d rays), %.1f s' % (what, m['t_pos_idx'].shape[0], stride, res, res,
                                                                                           f['covered'], rays, dt)}


# ---------------------------------------------------------------------------------------------------------------------
# rocprofv3 PMC passes of this very workload (rank 0, N = 1): counters cannot be read in-process, so the bench re-runs
# itself for a few steps under the profiler, one pass per counter group (separate --pmc passes, kernel trace only: no
# sys/hip/hsa trace domains), and reads the per-dispatch sums out of the rocpd database.

PMC_PASSES = [
    ['SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY',
     'SQ_WAVES', 'GRBM_GUI_ACTIVE'],
    ['FETCH_SIZE'],
    ['TCC_REQ_sum', 'WRITE_SIZE', 'TCC_MISS_sum'],
    # the instruction mix of the traversal loop: how many issue slots go to scalar / branch / memory instructions beside VALU
    ['SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAIT_ANY'],
]


PER_XCD_CYCLE_COUNTERS = ('GRBM_GUI_ACTIVE',)    # one row per XCD, each the cycle count of the whole dispatch: averaged, not summed


def _pmc_read(db_path, lead):
    """{kernel name: {counter: per-launch total}} and {kernel name: launches counted}.  A rocpd database holds several rows
    per (dispatch, counter) -- one per XCD / shader engine -- which are summed.  An env-shade launch issues its kernels once
    per chunk of the ray stream and the chunks behind the covered-pixel count are empty dispatches (~4 us): only dispatches
    whose `lead` counter reaches 10 % of the kernel's largest are averaged."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute('pragma table_info(pmc_events)')]
    name_col = 'counter_name' if 'counter_name' in cols else ('name' if 'name' in cols else cols[0])
    val_col = 'value' if 'value' in cols else ('counter_value' if 'counter_value' in cols else cols[-1])
    q = ('select k.name, p.dispatch_id, p.%s, sum(p.%s), count(*) from pmc_events p join kernels k '
         'on k.dispatch_id = p.dispatch_id group by k.name, p.dispatch_id, p.%s' % (name_col, val_col, name_col))
    per = {}
    for name, did, ctr, total, rows in db.execute(q).fetchall():
        per.setdefault(name, {}).setdefault(did, {})[ctr] = total / rows if ctr in PER_XCD_CYCLE_COUNTERS else total
    out, disp = {}, {}
    for name, dd in per.items():
        top = max((c.get(lead, 0.0) for c in dd.values()), default=0.0)
        real = [c for c in dd.values() if c.get(lead, 0.0) >= 0.1 * top] if top > 0 else list(dd.values())
        disp[name] = len(real)
        keys = set().union(*[set(c) for c in real]) if real else set()
        out[name] = {k: sum(c.get(k, 0.0) for c in real) / len(real) for k in keys}
    return out, disp


def collect_pmc(args, keep_dir=None, config=None, passes=None):
    """Run the PMC passes; returns (counters per kernel, note) -- counters is None when rocprofv3 is unavailable or failed.
    config / passes: another preset (the large-mesh object runs `dmtet800` with the two memory passes only)."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    merged, notes = {}, []
    own = config is None
    config = config or args.config
    passes = passes or PMC_PASSES
    child = [sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-child', '--config', config, '--steps', '2', '--warmup', '1',
             '--scaling', args.scaling, '--lock-pos', args.lock_pos, '--material-set', args.material_set]
    if args.tex_res is not None:
        child += ['--tex-res', str(args.tex_res)]
    for flag, v in (('--res', args.res), ('--n-samples-x', args.n_samples_x), ('--mesh', args.mesh), ('--subdiv', args.subdiv), ('--batch', args.batch)):
        if v is not None and own:
            child += [flag, str(v)]
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    for i, group in enumerate(passes):
        d = tempfile.mkdtemp(prefix='nvdr_pmc%d_' % i, dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc'] + group + ['-d', d, '-o', 'r', '--'] + child
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=args.pmc_timeout)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith('_results.db')]
            if r.returncode != 0 or not dbs:
                notes.append('pass %d (%s) failed rc=%d: %s' % (i, ' '.join(group), r.returncode, (r.stderr or r.stdout)[-300:]))
                continue
            ctrs, disp = _pmc_read(dbs[0], group[0])
            for kname, c in ctrs.items():
                merged.setdefault(kname, {}).update(c)
                merged[kname]['dispatches_pass%d' % i] = disp.get(kname, 0)
        except subprocess.TimeoutExpired:
            notes.append('pass %d (%s) timed out after %d s' % (i, ' '.join(group), args.pmc_timeout))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if keep_dir and merged:
        # the raw per-launch counter sums behind the roofline object, as a small table (the rocpd databases are ~30 MB each)
        os.makedirs(keep_dir, exist_ok=True)
        with open(os.path.join(keep_dir, 'pmc_counters_%s.md' % config), 'w') as f:
            f.write('rocprofv3 --kernel-trace --pmc <group> -- python bench.py --pmc-child --config %s --steps 2 --warmup 1 (one pass per group: %s); '
                    'per-launch sums over the non-empty dispatches\n\n| kernel | counter | per launch | launches |\n|---|---|---|---|\n'
                    % (config, ' / '.join(' '.join(g) for g in passes)))
            for kname in sorted(merged):
                if not any(t in kname for t in ('env_', 'light_grad', 'bilateral', 'bvh_', 'gbuffer', 'image_loss', 'compact')):
                    continue
                for ctr in sorted(merged[kname]):
                    if ctr.startswith('dispatches_pass'):
                        continue
                    n = max(merged[kname].get('dispatches_pass%d' % i, 0) for i in range(len(passes)))
                    f.write('| %s | %s | %.6g | %d |\n' % (kname[:90], ctr, merged[kname][ctr], n))
    return (merged or None), '; '.join(notes)


def find_kernel(counters, needle):
    """Counters of the first kernel whose name contains `needle` (or the first of several alternatives that occurs)."""
    for nd in ((needle,) if isinstance(needle, str) else needle):
        for name, c in counters.items():
            if nd in name:
                return c
    return None


def valu_figures(c, kernel_ms):
    """VALU-side figures of one kernel from its per-launch counter sums and its (un-profiled) duration."""
    insts, thread_cyc, active = c.get('SQ_INSTS_VALU'), c.get('SQ_THREAD_CYCLES_VALU'), c.get('SQ_ACTIVE_INST_VALU')
    if not (insts and thread_cyc and active and kernel_ms):
        return None
    lane_frac = thread_cyc / (64.0 * active)            # average share of the 64 lanes a VALU instruction executes for
    lane_ops = insts * 64.0 * lane_frac                 # active-lane VALU operations per launch
    achieved = lane_ops / (kernel_ms * 1e-3) / 1e12
    cycles = c.get('GRBM_GUI_ACTIVE')
    out = {'valu_wave_instructions': insts, 'active_lane_fraction': lane_frac, 'active_lane_ops': lane_ops,
           'achieved_Tlaneops': achieved, 'frac_of_lane_peak': achieved / VALU_PEAK_TLANEOPS,
           # issue slots: one wave64 VALU instruction per SIMD every 2 cycles (SIMD-32, MI355X_MICROARCH.md), lanes ignored
           'issue_frac_of_peak': insts * 2.0 / (N_CUS * SIMDS_PER_CU * CLOCK_GHZ * 1e9 * kernel_ms * 1e-3)}
    if cycles:
        out['profiled_kernel_cycles'] = cycles
        # SQ_ACTIVE_INST_VALU counts quad-cycles: share of the kernel's cycles a SIMD spends issuing VALU work, as the
        # counter block itself accounts it (4 cycles per instruction) -- the "VALUBusy" of the profiler
        out['valu_busy_counter'] = active * 4.0 / (N_CUS * SIMDS_PER_CU * cycles)
    if c.get('SQ_WAVE_CYCLES') and c.get('SQ_WAIT_INST_ANY'):
        out['wave_time_waiting_on_issue_or_memory'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
    mix = {k: c.get(k) for k in ('SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR') if c.get(k) is not None}
    if mix:
        # wave-instructions of every kind per SIMD and cycle: the traversal loop spends about as many issue slots on scalar
        # (exec-mask / branch) instructions as on VALU ones
        total = insts + sum(mix.values())
        out['instruction_mix_per_launch'] = dict(mix, SQ_INSTS_VALU=insts)
        out['valu_share_of_instructions'] = insts / total
        out['instructions_per_cycle_per_simd'] = total / (N_CUS * SIMDS_PER_CU * CLOCK_GHZ * 1e9 * kernel_ms * 1e-3)
    return out


def mem_figures(c, kernel_ms):
    out = {}
    if c.get('FETCH_SIZE') is not None and c.get('WRITE_SIZE') is not None:
        # gfx950: FETCH_SIZE tallies the 128-B requests of wide reads at 64 B (MI355X_MICROARCH.md, HBM): doubled. KB units.
        out['hbm_bytes'] = (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0
        out['fetch_bytes_corrected'] = 2.0 * c['FETCH_SIZE'] * 1024.0
        out['write_bytes'] = c['WRITE_SIZE'] * 1024.0
        if kernel_ms:
            out['hbm_GBs'] = out['hbm_bytes'] / (kernel_ms * 1e-3) / 1e9
            out['hbm_frac'] = out['hbm_GBs'] / HBM_PEAK_GBS
    if c.get('TCC_REQ_sum'):
        out['l2_requests'] = c['TCC_REQ_sum']
        if c.get('TCC_MISS_sum') is not None:
            out['l2_hit'] = 1.0 - c['TCC_MISS_sum'] / c['TCC_REQ_sum']
        if kernel_ms:
            out['l2_GBs_at_64B_per_request'] = c['TCC_REQ_sum'] * 64.0 / (kernel_ms * 1e-3) / 1e9
            out['l2_frac'] = out['l2_GBs_at_64B_per_request'] / L2_PEAK_GBS
    return out


def cpu_baseline_torch():
    """The literal baseline BASELINE.json configs[0] sketches, in full: bob 128x128, n_samples_x = 2 (4 spp, 8 shadow rays per covered
    pixel and pass), the shadow test as brute-force PyTorch-CPU ops over all 10 688 triangles (oracle/torch_baseline.py: torch.set_num_threads
    = the host's cores), forward + re-tracing backward, around the restated raygen program."""
    import torch
    from oracle import oracle as orc, scene_cpu, torch_baseline as tb
    nt = orc.max_threads()
    inp = scene_cpu.make_inputs('bob', 128, 128, 2, n_threads=nt)
    kw = scene_cpu.shade_kwargs(inp)
    g = torch.Generator().manual_seed(0)
    dg, sg = torch.rand(1, 128, 128, 3, generator=g), torch.rand(1, 128, 128, 3, generator=g)
    f, b, t = tb.direct_lighting_torch_shadow(inp['mesh'], kw, 2, diff_grad=dg, spec_grad=sg, n_threads=nt)
    rays = 2 * t['rays_per_pass']
    return {'value': rays / t['total_s'], 'unit': 'rays/s', 'cores': t['threads'], 'host_cores': nt, 'kind': 'port',
            'sample': 'BASELINE configs[0] in full: bob 128x128, n_samples_x=2, %d covered pixels, %d shadow rays (forward + re-traced backward) against '
                      '%d triangles by chunked torch ops on the CPU (%.1f s of the %.1f s; the rest is the restated raygen / shading program)'
                      % (t['covered'], rays, inp['mesh']['t_pos_idx'].shape[0], t['torch_shadow_fwd_s'] + t['bwd_s'], t['total_s']),
            'seconds': t['total_s']}


def other_config_object(name, args, dev):
    """One of the other BASELINE configs on the same line (rank 0, N = 1): a few timed iterations, rays from the counting launch; no
    counters (the dedicated `--config <name>` run has them)."""
    import torch
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    from nvdiffrecmc_amd import optixutils as ou
    pre = PRESETS[name]
    t0 = time.perf_counter()
    H, n, nv = pre['res'], pre['n'], pre['batch']
    lock = pre.get('lock_pos', True) or args.material_set != 'full'
    step = make_step(pre, args, dev, list(range(nv)), nv, lock, retrace_backward=True)
    for _ in range(4):
        step.step(1)
    K = 6
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for _ in range(K):
        step.step(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - w0
    with torch.no_grad():
        L = step.light
        _, ro, _, nrm, _, kd, ks = step.shade_inputs()
        P, n_box, n_tri, n_traced = ou.ops.env_shade_traversal_counts(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks, L.base, L._pdf,
                                                                     L.rows[:, 0], L.cols, n_samples_x=n, rnd_seed=0)
    out = {'workload': pre['what'] + ', batch of %d views' % nv, 'mesh_triangles': int(step.mesh['t_pos_idx'].shape[0]), 'covered_pixels': P,
           'rays_traversed_per_pass': n_traced, 'steps': K, 'ms_per_step': dt / K * 1e3, 'rays_per_sec': 2.0 * n_traced * K / dt,
           'geometry': 'locked' if lock else 'trained', 'trained_parameters': list(step.param_names), 'seconds': time.perf_counter() - t0}
    if not lock:
        out['geometry_note'] = GEOMETRY_NOTE
    del step
    torch.cuda.empty_cache()
    return out


def adam_object(step):
    """The parameter update alone, as the iteration runs it (tile-sparse textures: tiles without gradient and without history are skipped) and
    DENSE (every texel read and written: what the reference's mip-mapped textures, whose gradient reaches every texel, would cost): HIP-event
    medians of the one launch.  Run at the very end: it moves the parameters by a few zero-gradient steps."""
    import torch
    opt = step.opt
    if not hasattr(opt, 'active'):
        return None
    for p in step.params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    out = {}
    saved = list(opt.active)
    for tag in ('sparse', 'dense'):
        if tag == 'dense':
            opt.active = [None] * len(saved)
        ms = []
        for it in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            opt.step()
            b.record()
            b.synchronize()
            if it >= 2:
                ms.append(a.elapsed_time(b))
        out[tag + '_ms'] = statistics.median(ms)
    opt.active = saved
    out['parameter_bytes'] = int(sum(p.numel() for p in step.params) * 4)
    out['tiles_with_history'] = [int(a.sum().item()) if a is not None else None for a in saved]
    return out


def init_world1(dev):
    """A one-rank RCCL process group on this GPU (the only N a one-GPU box offers): the several-rank schedule then runs its real
    collectives -- a one-rank all-reduce moves nothing over xGMI, what it shows is the fixed cost of the path."""
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(_free_port()))
    with _stdout_to_stderr():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    return True


def exchange_object(step, ms_exposed):
    """config.exchange of a line that ran the several-rank schedule: what the last round sent and how long the main stream stood still for it."""
    rep = step._ex.report()
    rep['pipelined'] = bool(step.pipeline and len(step._ex_chunks) > 1)
    rep['exposed_ms'] = statistics.median(ms_exposed) if ms_exposed else None
    s1 = step.stage1_ms()
    rep['geometry_stage_ms'] = statistics.median(s1) if s1 else None      # main-stream time of the next iteration's geometry stage: the texture chunk's reduce runs under it
    rep['exposed_ms_note'] = ('median per iteration of the time the main stream waits on the exchange (HIP events around every wait: collectives not yet '
                              'finished + the scatter of the reduced tiles); the rest of the exchange runs under the next iteration\'s geometry stage')
    if getattr(step, '_union_views', None):
        rep['union_emulated_views'] = len(step._union_views)
    return rep


def one_view_object(args, dev, preset_name, eight_view_ms, lock=None):
    """The per-GPU share of the 8-GPU run on THIS box (rank 0, N = 1): one view of the batch in HIP graphs under the several-rank schedule
    (chunks ordered by the next iteration's need, tile-sparse texture chunk, pipelined with the next geometry stage) with a one-rank RCCL
    group doing the real collectives on the bytes an 8-rank run would send (the tile flags are OR-ed with the tiles ALL eight views touch).
    No xGMI time is in it -- `projected` prices the wire separately."""
    import torch
    pre = dict(PRESETS[preset_name])
    t0 = time.perf_counter()
    lock = pre.get('lock_pos', True) if lock is None else lock
    own = init_world1(dev)
    out = {'preset': preset_name, 'geometry': 'locked' if lock else 'trained'}
    try:
        for mode in ('sparse', 'dense'):
            step = make_step(pre, args, dev, [0], pre['batch'], lock, retrace_backward=True, use_graph=True, force_exchange=True, exchange_mode=mode,
                             union_views=list(range(pre['batch'])))
            for _ in range(12):
                step.step(1)
            step.measure_exposed = True
            K = 40
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            for _ in range(K):
                step.step(1)
            step.finish()
            torch.cuda.synchronize()
            dt = time.perf_counter() - w0
            ex = exchange_object(step, step.exposed_ms())
            out[mode] = {'ms_per_step': dt / K * 1e3, 'hip_graph': step._graphs is not None, 'exchange': ex}
            del step
            torch.cuda.empty_cache()
        # the wire, priced: ring all-reduce of S bytes over 8 GPUs moves 2 * 7/8 * S per GPU; bus bandwidth 150-250 GB/s assumed (xGMI: 7 links x ~153 GB/s
        # peak per GPU, MI355X_MICROARCH.md); latency floor 30 us per collective.  Only what is NOT hidden under the next geometry stage counts.
        proj = {}
        for mode in ('sparse', 'dense'):
            e = out[mode]['exchange']
            early = e['chunk_bytes_dense'][0] if len(e['chunk_bytes_dense']) > 1 else 0
            tex = e['bytes_sent'] - early
            wire = lambda nbytes, bw: 30e-6 + 2.0 * 7.0 / 8.0 * nbytes / (bw * 1e9)
            # the early chunk's wire time is exposed; the texture chunk's only where it outlasts the geometry stage it runs under (measured here, on the main stream)
            s1 = (e.get('geometry_stage_ms') or 0.0) * 1e-3
            lo, hi = [out[mode]['ms_per_step'] + 1e3 * (wire(early, bw) + max(0.0, wire(tex, bw) - s1)) for bw in (250.0, 150.0)]
            proj[mode] = {'ms_per_step_with_exposed_wire': [lo, hi], 'speedup_vs_8_views_on_one_gpu': [eight_view_ms / hi, eight_view_ms / lo]}
        out['projected_8gpu'] = dict(proj, eight_views_one_gpu_ms=eight_view_ms,
                                     note='8 views on one GPU / (one view under the several-rank schedule measured here, one-rank RCCL collectives included, + the ring '
                                          'all-reduce wire time at 150-250 GB/s bus bandwidth that the pipelined geometry stage does not cover); a projection -- the measured '
                                          'scaling curve is the driver\'s SCALE file when an 8-GPU node exists')
    finally:
        if own:
            import torch.distributed as dist
            with _stdout_to_stderr():
                dist.destroy_process_group()
    out['seconds'] = time.perf_counter() - t0
    return out


def large_mesh_object(args, dev):
    """The L2-spilling workload on the SAME bench line (rank 0, N = 1): `dmtet800` -- bob subdivided three times, 684 032 triangles
    (9 MB of eight-wide nodes + 33 MB of triangle records against 4 MB of L2 per XCD), 800x800, 64 spp, 8 views -- a few timed
    iterations, the traversal kernel's HIP-event time and ray count, and two memory-side PMC passes (FETCH_SIZE; TCC_REQ /
    WRITE_SIZE / TCC_MISS).  This is where "fraction of the HBM roofline" is a physical question (on bob the tree is L2 resident)."""
    import torch
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    from nvdiffrecmc_amd import optixutils as ou, renderutils as ru
    pre = PRESETS['dmtet800']
    t0 = time.perf_counter()
    H, n, nv = pre['res'], pre['n'], pre['batch']
    step = make_step(pre, args, dev, list(range(nv)), nv, not (args.material_set == 'full' and not pre.get('lock_pos', True) and args.lock_pos != 'on'),
                     retrace_backward=True)
    step_unlocked = step.optimize_geometry
    for _ in range(4):
        step.step(1)
    step.ctx.set_profiling(True)
    K = 8
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step.step(1)
        b.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - w0
    ms = [a.elapsed_time(b) for a, b in ev]
    n_f, (gen_ms, trace_ms, shade_ms) = step.ctx.stage_times(backward=False)
    step.ctx.set_profiling(False)
    with torch.no_grad():
        m = step.mask[..., None]
        _, ro, _, nrm, _, kd, ks = step.shade_inputs()
        L = step.light
        P, n_box, n_tri, n_traced = ou.ops.env_shade_traversal_counts(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks, L.base, L._pdf,
                                                                     L.rows[:, 0], L.cols, n_samples_x=n, rnd_seed=0)
        steps_per_ray = ou.ops.env_shade_traversal_counts.node_steps / max(n_traced, 1)
    n_tris = int(step.mesh['t_pos_idx'].shape[0])
    med = statistics.median(ms)
    out = {'preset': 'dmtet800', 'workload': pre['what'] + ', batch of %d views' % nv, 'mesh_triangles': n_tris, 'covered_pixels': P,
           'rays_traversed_per_pass': n_traced, 'steps': K, 'ms_per_step': dt / K * 1e3, 'median_ms_per_step': med,
           'steps_over_twice_the_median': sum(1 for v in ms if v > 2.0 * med),
           'rays_per_sec': 2.0 * n_traced * K / dt, 'kernel': DOMINANT, 'kernel_ms_hip_events': trace_ms, 'launches_timed': n_f,
           'kernel_rays_per_sec': n_traced / (trace_ms * 1e-3), 'node_steps_per_ray': steps_per_ray, 'box_tests_per_ray': n_box / max(n_traced, 1),
           'triangle_tests_per_ray': n_tri / max(n_traced, 1),
           'tree_bytes': {'oct_nodes_64B': None, 'triangle_records_48B': 48 * n_tris}}
    try:
        out['tree_bytes']['oct_nodes_64B'] = 64 * int(step.ctx.bvh_export_oct()[2]['nodes'])
    except Exception:
        pass
    del step
    torch.cuda.empty_cache()
    if not args.no_pmc:
        counters, note = collect_pmc(args, keep_dir=args.pmc_keep, config='dmtet800', passes=[['FETCH_SIZE'], ['TCC_REQ_sum', 'WRITE_SIZE', 'TCC_MISS_sum']])
        c = find_kernel(counters, DOMINANT) if counters else None
        if c:
            # per DISPATCH -> per pass: a ray stream larger than the context's byte budget (8 GiB by default) is cut into chunks, one dispatch
            # each, and the child runs 7 env-shade passes (see the roofline object below)
            chunks = max(1, int(round(c.get('dispatches_pass0', 7) / 7.0)))
            if chunks > 1:
                c = {k: (v_ * chunks if not k.startswith('dispatches_pass') else v_) for k, v_ in c.items()}
                out['dispatches_per_launch'] = chunks
            mem = mem_figures(c, trace_ms)
            out['hbm'] = {k: mem[k] for k in ('hbm_bytes', 'fetch_bytes_corrected', 'write_bytes', 'hbm_GBs', 'hbm_frac') if k in mem}
            out['hbm']['note'] = ('L2-miss (fabric) traffic / HBM peak: the working set (9 MB of nodes + 33 MB of triangle records) fits the 256 MB Infinity '
                                  'Cache, whose hits FETCH_SIZE counts.  2 x FETCH_SIZE is calibrated for this access pattern (profiles/r04_gather64_calibration.md: '
                                  'a divergent 64-byte gather that misses L2 moves one 128-byte line and is tallied at 64 B); a pure gather kernel reaches 7.9 TB/s '
                                  'of line traffic on a 42 MB array, 6.9 TB/s on 1 GB')
            out['l2'] = {k: mem[k] for k in ('l2_requests', 'l2_hit', 'l2_GBs_at_64B_per_request', 'l2_frac') if k in mem}
        if note:
            out['pmc_note'] = note
    out['geometry'] = 'trained (v_pos, lr %g)' % BENCH_LR_POS if step_unlocked else 'locked'
    if step_unlocked:
        out['geometry_note'] = GEOMETRY_NOTE
    out['seconds'] = time.perf_counter() - t0
    return out


# ---------------------------------------------------------------------------------------------------------------------

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
    ap.add_argument('--exchange', choices=('auto', 'dense', 'sparse'), default='auto',
                    help='gradient exchange of the texture chunk at N > 1: dense = the whole bucket, sparse / auto = the tiles some rank touched (falls back to dense by itself)')
    ap.add_argument('--no-pipeline', action='store_true', help='N > 1: wait for the texture chunk inside the iteration instead of under the next geometry stage')
    ap.add_argument('--exchange-world1', action='store_true',
                    help='N = 1: run the several-rank schedule with a one-rank RCCL group (fixed cost of the exchange path; tile flags OR-ed with the tiles all views of the batch touch)')
    ap.add_argument('--no-one-view', action='store_true', help='skip the `one_view` object (per-GPU share of the 8-GPU run) of the default N = 1 line')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), default='auto',
                    help='capture the iteration in HIP graphs (auto: when a rank renders <= 2 views -- the launch-bound regime -- or there are several ranks)')
    return ap.parse_args()


class _stdout_to_stderr:
    """RCCL announces itself on STDOUT when it is loaded ("Librccl path : ..."); the one JSON line must stay the only thing there."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:        # the banner sits in the C library's stdout buffer (a pipe is fully buffered) and would come out at process exit
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawned(local_rank, world, port, args):
    os.environ.update({'RANK': str(local_rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    run(args)


def main():
    args = parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU fallback')
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
                'why_valu': 'rocprofv3 counters of this run: VALU issue dominates the kernel while its HBM and L2 fractions (hbm, l2 below) '
                            'are small -- divergent traversal of a tree that is cache resident; peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz.  '
                            'The fraction is the active-lane VALU rate, not a claim that the issue port is full: what sets the time is a wavefront\'s '
                            'dependent chain per node step times the 8 wavefronts a SIMD holds (DESIGN.md section 5, profiles/r05_trace_l1_bound.md)',
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
                                    'trained (v_pos, lr %g): BVH, vertex normals / tangents and the G-buffer rebuilt from the moving vertices every iteration; ' % BENCH_LR_POS + GEOMETRY_NOTE),
                       'allreduce_bytes_per_step': getattr(step, 'allreduce_bytes', 0)},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline and not preset['subdiv']:
            # bounded sample: ~4e10 ray-triangle tests (10-30 s on the box's host cores) = covered pixels x 4S rays x triangles
            stride = max(2, int(math.ceil(math.sqrt(0.23 * H * W * 4 * S * n_tris / 4e10))))
            out['cpu_baseline'] = cpu_baseline(preset['mesh'], H, n, 0, n_views, stride=stride)
        else:
            out['cpu_baseline'] = None   # N > 1, --no-cpu-baseline, or a subdivided mesh (brute force over 684 k triangles is not a bounded sample)
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
        if world == 1 and args.config == 'bob512' and not args.no_one_view and not args.pmc_child and args.res is None and args.subdiv is None and args.batch is None \
                and not forced:
            step = None
            torch.cuda.empty_cache()
            try:
                with _stdout_to_stderr():
                    out['config']['one_view'] = one_view_object(args, dev, 'bob512', dt / args.steps * 1e3)
            except Exception as e:
                out['config']['one_view'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and args.config == 'bob512' and not args.no_other_configs and not args.no_large_mesh and not args.pmc_child and args.res is None and args.subdiv is None \
                and args.batch is None:
            out['other_configs'] = {}
            step = None
            torch.cuda.empty_cache()
            for name in ('spot512x256', 'hotdog512x256'):
                try:
                    out['other_configs'][name] = other_config_object(name, args, dev)
                except Exception as e:
                    out['other_configs'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and args.config == 'bob512' and not args.no_large_mesh and not args.pmc_child and args.res is None and args.subdiv is None:
            try:
                step = None
                torch.cuda.empty_cache()
                out['large_mesh'] = large_mesh_object(args, dev)
            except Exception as e:      # never lose the headline line to the extra object
                out['large_mesh'] = {'error': '%s: %s' % (type(e).__name__, e)}
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
