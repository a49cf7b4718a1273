"""The sub-benchmarks behind the extra objects of bench.py's JSON line (presets, counter passes, CPU baselines, the one-view / large-mesh /
other-config objects).  bench.py itself holds the timed protocol of the driver's contract and reads top to bottom; everything here is
called from its `run()` on rank 0 after the timed region."""
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
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
    # round 6: meshes of the kind DMTet really produces (tools/make_dmtet_mesh.py: marching tets over the reference's tet grid with seeded SDFs --
    # irregular triangles, slivers, floaters), not the regular patches of a subdivided mesh
    'dmtet64_800': dict(mesh='dmtet64_mid', res=800, n=8, batch=8, subdiv=0, lock_pos=False, tex_res=1024,
                        metric='MC shadow rays/sec (fwd+bwd train iteration, 800x800 64spp, 140k-triangle marching-tets mesh)',
                        what='nerf_lego.json on a marching-tets extraction (tet grid tiled to the 128^3 class, seeded rough surface with floaters: '
                             '140 114 irregular triangles), 800x800, 64 spp (n_samples_x=8)'),
    'dmtet64_512x256': dict(mesh='dmtet64_mid', res=512, n=16, batch=8, subdiv=0, lock_pos=False, tex_res=1024,
                            metric='MC shadow rays/sec (fwd+bwd train iteration, 512x512 256spp, 140k-triangle marching-tets mesh)',
                            what='nerfactor_hotdog.json on the same marching-tets extraction (140 114 irregular triangles), 512x512, 256 spp (n_samples_x=16)'),
    'dmtet64_init512': dict(mesh='dmtet64_init', res=512, n=8, batch=8, subdiv=0, lock_pos=False, tex_res=1024,
                            metric='MC shadow rays/sec (fwd+bwd train iteration, 512x512 64spp, iteration 0 of a DMTet run)',
                            what='iteration 0 of a DMTet run at the default 64^3 grid: sdf = U[0,1) - 0.1 per grid vertex (geometry/dmtet.py:173), 77 705 triangles '
                                 'filling the whole grid volume, 512x512, 64 spp (n_samples_x=8)'),
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
                              rebuild_every=rebuild_every(args, lock_pos),
                              **dict(dict(exchange_mode=args.exchange, pipeline=not args.no_pipeline), **kw))


REBUILD_EVERY_TRAINED = 8


def rebuild_every(args, lock_pos):
    """BVH policy of a preset: locked geometry is rebuilt every iteration like the reference does (dlmesh.py:50); trained geometry rebuilds every
    8th iteration and refits in between (trainer.DirectLightingStep rebuild_every; --rebuild-every 1 = a rebuild every iteration)."""
    if lock_pos:
        return 1
    k = getattr(args, 'rebuild_every', None)
    return REBUILD_EVERY_TRAINED if k is None else max(1, int(k))


def bvh_policy_note(args, lock_pos):
    k = rebuild_every(args, lock_pos)
    if lock_pos or k == 1:
        return 'rebuilt every iteration (rebuild=1, as geometry/dlmesh.py:50)'
    return ('rebuilt every %d-th iteration, REFITTED in between (rebuild=0: the last rebuild\'s topology, new boxes, the eight-wide collapse redone; visibility and '
            'primary hits are exact with any valid tree -- tests/test_gpu_bvh.py, test_gpu_geometry.py); the reference passes rebuild=1 every iteration '
            '(geometry/dlmesh.py:50): --rebuild-every 1 reproduces that' % k)


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
    if getattr(args, 'rebuild_every', None) is not None:
        child += ['--rebuild-every', str(args.rebuild_every)]
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
            'cores_note': 'torch intra-op threads actually used: beyond ~32 the [256 rays x 10 688 triangles] element-wise ops only contend (all 128 threads of the box: 2.7 x slower, '
                          'oracle/torch_baseline.py); BASELINE.md asks for os.cpu_count() threads -- that would report a slower baseline',
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


def one_view_object(args, dev, preset_name, eight_view_ms, lock=None, modes=('dense', 'sparse')):
    """The per-GPU share of the 8-GPU run on THIS box (rank 0, N = 1): one view of the batch in HIP graphs, plain (`plain`) and under the
    several-rank schedule (chunks ordered by the next iteration's need, the texture chunk's reduce pipelined with the next iteration's
    geometry stage; exchange `dense` = the default, `sparse` = the tile-sparse texture chunk) with a one-rank RCCL group doing the real
    collectives on the bytes an 8-rank run would send (the tile flags are OR-ed with the tiles ALL eight views touch).
    No xGMI time is in it -- `projected_8gpu` prices the wire separately."""
    import torch
    os.environ.setdefault('NVDR_MESH_CACHE', os.path.join(tempfile.gettempdir(), 'nvdr_mesh_cache'))
    os.makedirs(os.environ['NVDR_MESH_CACHE'], exist_ok=True)
    pre = dict(PRESETS[preset_name])
    t0 = time.perf_counter()
    lock = pre.get('lock_pos', True) if lock is None else lock
    own = init_world1(dev)
    out = {'preset': preset_name, 'geometry': 'locked' if lock else 'trained', 'bvh': bvh_policy_note(args, lock)}
    K = 40
    try:
        if eight_view_ms is None:       # the batch of views on this one GPU, for the ratio (a few iterations)
            nv = pre['batch']
            step = make_step(pre, args, dev, list(range(nv)), nv, lock, retrace_backward=True)
            for _ in range(4):
                step.step(1)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            for _ in range(8):
                step.step(1)
            torch.cuda.synchronize()
            eight_view_ms = (time.perf_counter() - w0) / 8 * 1e3
            out['eight_views_measured_here'] = {'ms_per_step': eight_view_ms, 'steps': 8}
            del step
            torch.cuda.empty_cache()
        step = make_step(pre, args, dev, [0], pre['batch'], lock, retrace_backward=True, use_graph=True)
        for _ in range(12):
            step.step(1)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        for _ in range(K):
            step.step(1)
        torch.cuda.synchronize()
        out['plain'] = {'ms_per_step': (time.perf_counter() - w0) / K * 1e3, 'hip_graph': step._graphs is not None}
        del step
        torch.cuda.empty_cache()
        for mode in modes:
            step = make_step(pre, args, dev, [0], pre['batch'], lock, retrace_backward=True, use_graph=True, force_exchange=True, exchange_mode=mode,
                             union_views=list(range(pre['batch'])))
            for _ in range(12):
                step.step(1)
            step.measure_exposed = True
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            for _ in range(K):
                step.step(1)
            step.finish()
            torch.cuda.synchronize()
            dt = time.perf_counter() - w0
            ex = exchange_object(step, step.exposed_ms())
            out[mode] = {'ms_per_step': dt / K * 1e3, 'hip_graph': step._graphs is not None, 'exchange': ex,
                         'rebuild_on_side_stream': bool(getattr(step, 'split_stage2', False))}
            del step
            torch.cuda.empty_cache()
        # The wire, priced: a ring all-reduce of S bytes over 8 GPUs moves 2 * 7/8 * S per GPU; bus bandwidth 150-250 GB/s assumed (xGMI: 7 links x ~153
        # GB/s peak per GPU, MI355X_MICROARCH.md); latency floor 30 us per collective.  The early chunk's wire time is exposed in full; the texture
        # chunk's only where it outlasts the geometry stage the MAIN stream runs meanwhile (trained geometry: rebuild / refit, vertex frames, G-buffer).
        # With locked geometry the rebuild runs on a side stream beside the wire (and beside the next lookups / sample generation if it is not done by
        # then): the main stream has nothing to run under the wire, so all of the texture chunk's wire time is priced as exposed.
        proj = {}
        for mode in modes:
            e = out[mode]['exchange']
            early = e['chunk_bytes_dense'][0] if len(e['chunk_bytes_dense']) > 1 else 0
            tex = e['bytes_sent'] - early
            wire = lambda nbytes, bw: 30e-6 + 2.0 * 7.0 / 8.0 * nbytes / (bw * 1e9)
            s1 = (e.get('geometry_stage_ms') or 0.0) * 1e-3
            lo, hi = [out[mode]['ms_per_step'] + 1e3 * (wire(early, bw) + max(0.0, wire(tex, bw) - s1)) for bw in (250.0, 150.0)]
            proj[mode] = {'ms_per_step_with_exposed_wire': [lo, hi], 'speedup_vs_8_views_on_one_gpu': [eight_view_ms / hi, eight_view_ms / lo],
                          'wire_ms_early_chunk': [1e3 * wire(early, 250.0), 1e3 * wire(early, 150.0)], 'wire_ms_texture_chunk': [1e3 * wire(tex, 250.0), 1e3 * wire(tex, 150.0)]}
        out['projected_8gpu'] = dict(proj, eight_views_one_gpu_ms=eight_view_ms,
                                     note='8 views on one GPU / (one view under the several-rank schedule measured here, one-rank RCCL collectives included, + the ring '
                                          'all-reduce wire time at 150-250 GB/s bus bandwidth that the main stream has no geometry-stage work to run under); a projection -- '
                                          'the measured scaling curve is the driver\'s SCALE file when an 8-GPU node exists')
    finally:
        if own:
            import torch.distributed as dist
            with _stdout_to_stderr():
                dist.destroy_process_group()
    out['seconds'] = time.perf_counter() - t0
    return out


def flat_scaling_keys(prefix, ov, policy='dense'):
    """The one-view object's headline numbers as FLAT config keys (the driver's record keeps only flat keys of `config`): per-GPU share of the 8-GPU
    run plain and under the several-rank schedule, the main stream's wait on the exchange, the priced 8-GPU speed-up (lo = 150, hi = 250 GB/s)."""
    if not ov or 'error' in ov or policy not in ov:
        return {}
    sched, proj = ov[policy], (ov.get('projected_8gpu') or {}).get(policy) or {}
    sp = proj.get('speedup_vs_8_views_on_one_gpu') or [None, None]
    return {prefix + '_ms': ov['plain']['ms_per_step'], prefix + '_schedule_ms': sched['ms_per_step'],
            prefix + '_exposed_exchange_ms': (sched.get('exchange') or {}).get('exposed_ms'),
            prefix + '_exchange_mode': (sched.get('exchange') or {}).get('mode'),
            prefix + '_eight_views_ms': (ov.get('projected_8gpu') or {}).get('eight_views_one_gpu_ms'),
            prefix.replace('one_view', 'projected_8gpu') + '_speedup_lo': sp[0], prefix.replace('one_view', 'projected_8gpu') + '_speedup_hi': sp[1]}


def validation_object(args, dev):
    """The reference's VALIDATION sample count (train.py:263: n_samples = 32, S = 1024 strata, 2 048 shadow rays per covered pixel) on one 512^2 view of
    bob: the env-shade forward pass alone (validation renders no gradient), HIP-event stage times; parity of this path: tests/test_gpu_env_shade.py."""
    import torch
    from nvdiffrecmc_amd import optixutils as ou
    t0 = time.perf_counter()
    pre = dict(PRESETS['bob512'], n=32)
    step = make_step(pre, args, dev, [0], pre['batch'], True, retrace_backward=True)
    with torch.no_grad():
        _, ro, _, nrm, _, kd, ks = step.shade_inputs()
        L = step.light
        run = lambda seed: ou.optix_env_shade(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks, L.base, L._pdf, L.rows[:, 0], L.cols,
                                               BSDF='pbr', n_samples_x=32, rnd_seed=seed, shadow_scale=1.0)
        for it in range(3):
            run(it)
        step.ctx.set_profiling(True)
        K = 8
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for it, (a, b) in enumerate(ev):
            a.record()
            run(10 + it)
            b.record()
        torch.cuda.synchronize()
        n_f, (gen_ms, trace_ms, shade_ms) = step.ctx.stage_times(backward=False)
        step.ctx.set_profiling(False)
        ms = statistics.median(a.elapsed_time(b) for a, b in ev)
        P, n_box, n_tri, n_traced = ou.ops.env_shade_traversal_counts(step.ctx, step.mask, ro, step.gb_pos, nrm, step.view_pos, kd, ks, L.base, L._pdf,
                                                                     L.rows[:, 0], L.cols, n_samples_x=32, rnd_seed=0)
    out = {'workload': 'bob 512x512, ONE view, n_samples_x = 32 (train.py:263: the validation render; 1024 spp, 2 048 shadow rays per covered pixel), env-shade forward only',
           'covered_pixels': P, 'rays_traversed': n_traced, 'ms_per_forward': ms, 'gen_ms': gen_ms, 'trace_ms': trace_ms, 'shade_ms': shade_ms,
           'rays_per_sec': n_traced / (ms * 1e-3), 'trace_kernel_rays_per_sec': n_traced / (trace_ms * 1e-3), 'seconds': time.perf_counter() - t0}
    del step
    torch.cuda.empty_cache()
    return out


def large_mesh_object(args, dev, preset_name='dmtet64_800'):
    """A mesh whose tree does not fit the L2s on the SAME bench line (rank 0, N = 1): a few timed iterations, the traversal kernel's HIP-event time
    and ray count, and two memory-side PMC passes (FETCH_SIZE; TCC_REQ / WRITE_SIZE / TCC_MISS) -- where "fraction of the HBM roofline" is a physical
    question (on bob the tree is L2 resident).  `dmtet64_800` (the line's `large_mesh`, round 6): a marching-tets extraction, 140 114 irregular
    triangles -- what DMTet hands the BVH builder; `dmtet800` (`large_mesh_regular`, the `large_mesh` of rounds 3-5): bob subdivided three times,
    684 032 triangles in regular coplanar patches (9 MB of eight-wide nodes + 33 MB of triangle records against 4 MB of L2 per XCD)."""
    import torch
    from nvdiffrecmc_amd.trainer import DirectLightingStep
    from nvdiffrecmc_amd import optixutils as ou, renderutils as ru
    os.environ.setdefault('NVDR_MESH_CACHE', os.path.join(tempfile.gettempdir(), 'nvdr_mesh_cache'))
    os.makedirs(os.environ['NVDR_MESH_CACHE'], exist_ok=True)
    pre = PRESETS[preset_name]
    t0 = time.perf_counter()
    H, n, nv = pre['res'], pre['n'], pre['batch']
    lock = not (args.material_set == 'full' and not pre.get('lock_pos', True) and args.lock_pos != 'on')
    step = make_step(pre, args, dev, list(range(nv)), nv, lock, retrace_backward=True)
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
    out = {'preset': preset_name, 'workload': pre['what'] + ', batch of %d views' % nv, 'mesh_triangles': n_tris, 'covered_pixels': P,
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
        counters, note = collect_pmc(args, keep_dir=args.pmc_keep, config=preset_name, passes=[['FETCH_SIZE'], ['TCC_REQ_sum', 'WRITE_SIZE', 'TCC_MISS_sum']])
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
            out['hbm']['note'] = ('L2-miss (fabric) traffic / HBM peak: the working set (%.1f MB of nodes + %.1f MB of triangle records) fits the 256 MB Infinity '
                                  % ((out['tree_bytes']['oct_nodes_64B'] or 0) / 1e6, out['tree_bytes']['triangle_records_48B'] / 1e6) +
                                  'Cache, whose hits FETCH_SIZE counts.  2 x FETCH_SIZE is calibrated for this access pattern (profiles/r04_gather64_calibration.md: '
                                  'a divergent 64-byte gather that misses L2 moves one 128-byte line and is tallied at 64 B); a pure gather kernel reaches 7.9 TB/s '
                                  'of line traffic on a 42 MB array, 6.9 TB/s on 1 GB')
            out['l2'] = {k: mem[k] for k in ('l2_requests', 'l2_hit', 'l2_GBs_at_64B_per_request', 'l2_frac') if k in mem}
        if note:
            out['pmc_note'] = note
    out['geometry'] = 'trained (v_pos, lr %g)' % BENCH_LR_POS if step_unlocked else 'locked'
    out['bvh'] = bvh_policy_note(args, not step_unlocked)
    if step_unlocked:
        out['geometry_note'] = GEOMETRY_NOTE
    out['seconds'] = time.perf_counter() - t0
    return out


def extras_child(args):
    """Child process of the default N = 1 line (bench.py --extras-child <8-view ms>): the extra objects, most important first, each printed as
    ONE line `EXTRA <key> <json>` the moment it is done."""
    import torch
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    eight_ms = float(args.extras_child)

    def emit(key, fn, *a, quiet=False, **k):
        torch.cuda.empty_cache()
        try:
            if quiet:
                with _stdout_to_stderr():
                    val = fn(*a, **k)
            else:
                val = fn(*a, **k)
        except Exception as e:
            val = {'error': '%s: %s' % (type(e).__name__, e)}
        print('EXTRA %s %s' % (key, json.dumps(val)), flush=True)
        return val

    if not args.no_one_view:
        emit('one_view', one_view_object, args, dev, 'bob512', eight_ms, quiet=True)
    reg = None
    if not args.no_large_mesh:
        emit('large_mesh', large_mesh_object, args, dev, 'dmtet64_800')
        reg = emit('large_mesh_regular', large_mesh_object, args, dev, 'dmtet800')
        if not args.no_one_view:
            emit('large_mesh_regular/one_view_locked', one_view_object, args, dev, 'dmtet800', None, lock=True, modes=('dense',), quiet=True)
            emit('large_mesh_regular/one_view_trained', one_view_object, args, dev, 'dmtet800', (reg or {}).get('ms_per_step'), lock=False, modes=('dense',), quiet=True)
    if not args.no_validation:
        emit('validation_n32', validation_object, args, dev)
    if not args.no_other_configs and not args.no_large_mesh:
        for name in ('spot512x256', 'hotdog512x256'):
            emit('other_configs/' + name, other_config_object, name, args, dev)


def collect_extras(args, eight_ms):
    """Run extras_child in a subprocess and gather its `EXTRA` lines until it exits or --extras-timeout expires (then the child -- this exact
    process -- is killed and what has arrived is kept).  Returns ([(key, value)], note)."""
    import threading
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--extras-child', repr(float(eight_ms)), '--lock-pos', args.lock_pos, '--material-set', args.material_set,
           '--exchange', args.exchange, '--pmc-timeout', str(args.pmc_timeout)]
    for flag, on in (('--no-one-view', args.no_one_view), ('--no-large-mesh', args.no_large_mesh), ('--no-validation', args.no_validation),
                     ('--no-other-configs', args.no_other_configs), ('--no-pmc', args.no_pmc), ('--no-pipeline', args.no_pipeline)):
        if on:
            cmd.append(flag)
    for flag, v in (('--tex-res', args.tex_res), ('--rebuild-every', args.rebuild_every), ('--pmc-keep', args.pmc_keep)):
        if v is not None:
            cmd += [flag, str(v)]
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, env=env, cwd=ROOT)
    got = []

    def reader():
        for line in proc.stdout:
            if line.startswith('EXTRA '):
                try:
                    _, key, payload = line.split(' ', 2)
                    got.append((key, json.loads(payload)))
                except Exception:
                    pass
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    note = None
    try:
        proc.wait(timeout=args.extras_timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
        note = 'the extra objects were cut off after %d s (--extras-timeout); %d arrived: %s' % (args.extras_timeout, len(got), ', '.join(k for k, _ in got))
    th.join(timeout=5)
    if note is None and proc.returncode != 0:
        note = 'the child process of the extra objects exited with code %s after %d object(s)' % (proc.returncode, len(got))
    if note is None and not got:
        note = 'no extra object arrived (%.0f s)' % (time.perf_counter() - t0)
    return got, note


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
