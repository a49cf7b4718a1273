"""Host-side logic that needs no GPU: scene generators, API surface parity with the reference modules, the
opt-in torch formulations behind use_python=True."""
import inspect
import math
import os

import numpy as np
import pytest
import torch

from nvdiffrecmc_amd import scene as sc
from oracle import renderutils_ref as rr
from tests.util import assert_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_matches_dataset_mesh_convention():
    mv, mvp, campos = sc.camera(2, 8)
    assert abs(campos.norm().item() - 3.0) < 1e-5                  # radius 3 orbit (train.py:42)
    p = sc.perspective()
    assert p[1, 1].item() == pytest.approx(-1 / math.tan(math.radians(45) / 2))
    assert torch.allclose(mvp, p @ mv)
    ro, rd = sc.primary_rays(mv, 8, 8)
    assert torch.allclose(ro[0, 0], campos, atol=1e-6) and torch.allclose(rd.norm(dim=-1), torch.ones(8, 8), atol=1e-6)
    # the central ray looks at the origin
    c = sc.primary_rays(mv, 1, 1)[1][0, 0]
    assert torch.allclose(c, -campos / campos.norm(), atol=1e-6)


def test_perms_table_rows_are_permutations_and_seeded():
    p = sc.perms_table(3, seed=0, n_perms=64)
    assert p.dtype == torch.int32 and p.shape == (64, 9)
    assert torch.equal(torch.sort(p, dim=-1)[0], torch.arange(9, dtype=torch.int32).expand(64, 9))
    assert torch.equal(p, sc.perms_table(3, seed=0, n_perms=64)) and not torch.equal(p, sc.perms_table(3, seed=1, n_perms=64))


def test_meshes_and_subdivision():
    m = sc.load_mesh('bob')
    assert m['v_pos'].shape == (5344, 3) and m['t_pos_idx'].shape == (10688, 3) and m['t_pos_idx'].dtype == torch.int32
    assert torch.allclose(m['v_nrm'].norm(dim=-1), torch.ones(5344), atol=1e-5)
    v, t = sc.subdivide(m['v_pos'], m['t_pos_idx'], 1)
    assert t.shape[0] == 4 * 10688 and v.shape[0] == 5344 + 16032     # V + E for a closed genus-1... (E = 3T/2)
    assert int(t.max()) == v.shape[0] - 1
    s = sc.load_mesh('spot')
    assert s['t_pos_idx'].shape == (5856, 3) and tuple(s['ks'].tolist()) == pytest.approx((0.0, 0.2, 1.0))


def test_env_maps_and_tables():
    assert torch.equal(sc.env_map('E0', 16), torch.full((16, 16, 3), 0.5))
    e = sc.env_map('E1', 64)
    assert e.min().item() >= 1e-4 and e.max().item() > 100 and torch.equal(e, sc.env_map('E1', 64))
    pdf, rows, cols = sc.light_tables(e)
    assert rows.shape == (64, 64) and torch.equal(rows[:, 0], rows[:, 5])


def test_public_api_surface_matches_reference_modules():
    import nvdiffrecmc_amd.optixutils as ou
    import nvdiffrecmc_amd.renderutils as ru
    assert ou.__all__ == ["OptiXContext", "optix_build_bvh", "optix_env_shade", "bilateral_denoiser"]
    assert list(inspect.signature(ou.optix_env_shade).parameters) == [
        'optix_ctx', 'mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks', 'light', 'pdf', 'rows', 'cols',
        'BSDF', 'n_samples_x', 'rnd_seed', 'shadow_scale']
    d = {k: v.default for k, v in inspect.signature(ou.optix_env_shade).parameters.items()}
    assert (d['BSDF'], d['n_samples_x'], d['rnd_seed'], d['shadow_scale']) == ('pbr', 8, None, 1.0)
    assert list(inspect.signature(ou.optix_build_bvh).parameters) == ['optix_ctx', 'verts', 'tris', 'rebuild']
    assert list(inspect.signature(ou.bilateral_denoiser).parameters) == ['col', 'nrm', 'zdz', 'sigma']
    assert sorted(ru.__all__) == sorted(["xfm_vectors", "xfm_points", "image_loss", "prepare_shading_normal", "lambert",
                                         "frostbite_diffuse", "pbr_specular", "pbr_bsdf", "_fresnel_shlick", "_ndf_ggx",
                                         "_lambda_ggx", "_masking_smith"])
    sig = inspect.signature(ru.pbr_bsdf).parameters
    assert list(sig) == ['kd', 'arm', 'pos', 'nrm', 'view_pos', 'light_pos', 'min_roughness', 'bsdf', 'use_python']
    assert sig['min_roughness'].default == 0.08 and sig['bsdf'].default == 'lambert'
    sig = inspect.signature(ru.image_loss).parameters
    assert (sig['loss'].default, sig['tonemapper'].default, sig['use_python'].default) == ('l1', 'none', False)
    sig = inspect.signature(ru.prepare_shading_normal).parameters
    assert (sig['two_sided_shading'].default, sig['opengl'].default) == (True, True)
    from nvdiffrecmc_amd.denoiser import BilateralDenoiser
    d = BilateralDenoiser(0.5)
    assert d.sigma == 1.0 and d.N == 2 * math.ceil(2.5) + 1


def test_use_python_formulations_agree_with_oracle():
    """The opt-in torch path of the API (use_python=True) against the oracle restatement, on the CPU."""
    import nvdiffrecmc_amd.renderutils as ru
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.rand(*s, generator=g)
    n = lambda: torch.nn.functional.normalize(R(1, 6, 5, 3), dim=-1)
    a = [R(1, 6, 5, 3), R(1, 6, 5, 3), R(1, 6, 5, 3), n(), R(1, 1, 1, 3) + 2, R(1, 1, 1, 3) + 2]
    for b in ('lambert', 'frostbite'):
        assert_close(ru.pbr_bsdf(*a, bsdf=b, use_python=True), rr.pbr_bsdf(*a, bsdf=b), 5e-5)
    assert_close(ru.prepare_shading_normal(a[0], a[4], None, a[3], n(), n(), use_python=True).shape, (1, 6, 5, 3), 0)
    pn = R(1, 6, 5, 3)
    args = (a[0], a[4], pn, a[3], n(), n())
    assert_close(ru.prepare_shading_normal(*args, use_python=True), rr.prepare_shading_normal(*args), 2e-5)
    for loss in ('l1', 'mse', 'smape', 'relmse', 'n2n'):
        for tm in ('none', 'log_srgb'):
            assert_close(ru.image_loss(a[0], a[1], loss, tm, use_python=True), rr.image_loss(a[0], a[1], loss, tm), 1e-5)
    p, m = R(1, 7, 3), R(2, 4, 4)
    assert_close(ru.xfm_points(p, m, use_python=True), rr.xfm_points(p, m), 1e-6)
    assert ru.xfm_vectors(p, m, use_python=True).shape == (2, 7, 3)


def test_bench_algorithmic_byte_formula():
    import bench
    # 10 BVH2 node visits (32 B each, the reference accounting layout) + 2 triangle tests (36 B) by 90 000 traversed rays
    total, trace, trav = bench.algorithmic_bytes(1, 512, 512, 1000, 64, 256, bvh2_nodes=10, bvh2_tris=2, n_traced=90000)
    assert trav == 32 * 10 + 36 * 2
    assert total == trav + (4 * 512 * 512 + 60 * 1000 + 24 * 512 * 512) + 1000 * 64 * 128   # 128 B per stratum at 256^2
    assert trace == trav + 21 * 90000 + 16 * 1000       # per traversed ray: 16 B direction + 4 B list entry + 1 B visibility
    # the roofline object never reports a fraction above 1 of a real ceiling: VALU lane peak = 256 x 4 x 32 x 2.4 GHz
    assert abs(bench.VALU_PEAK_TLANEOPS - 78.6432) < 1e-3
    v = bench.valu_figures({'SQ_INSTS_VALU': 2.29e9, 'SQ_THREAD_CYCLES_VALU': 64 * 0.58 * 2.297e9, 'SQ_ACTIVE_INST_VALU': 2.297e9,
                            'GRBM_GUI_ACTIVE': 10.5e6}, 4.5)
    assert 0.0 < v['frac_of_lane_peak'] < 1.0 and abs(v['active_lane_fraction'] - 0.58) < 1e-6
    m = bench.mem_figures({'FETCH_SIZE': 737442.3, 'WRITE_SIZE': 224857.5, 'TCC_REQ_sum': 1.88e8, 'TCC_MISS_sum': 1.2e7}, 4.5)
    assert abs(m['hbm_bytes'] - (2 * 737442.3 + 224857.5) * 1024) < 1 and 0 < m['hbm_frac'] < 1 and 0 < m['l2_frac'] < 1
    # the shading kernels have two names (64 spp launches run the queue kernels): the first alternative that occurs is taken
    counters = {'void env_shade_kernel<true>(ShadeParams)': {'x': 1}, 'void env_shade_queue_kernel<false>(ShadeParams)': {'x': 2},
                'void env_shade_kernel<false>(ShadeParams)': {'x': 3}}
    assert bench.find_kernel(counters, ('env_shade_queue_kernel<true', 'env_shade_kernel<true'))['x'] == 1
    assert bench.find_kernel(counters, ('env_shade_queue_kernel<false', 'env_shade_kernel<false'))['x'] == 2
    assert bench.find_kernel(counters, 'env_gen_kernel') is None


def test_rocpd_tools_on_a_synthetic_database(tmp_path):
    """tools/rocpd_summary.py (kernel table + roctx ranges) and tools/rocpd_iteration.py (timeline of one iteration) against a small
    database with the columns of rocprofv3's `kernels` and `regions` views."""
    import sqlite3
    import subprocess
    import sys
    dbp = str(tmp_path / 'r.db')
    db = sqlite3.connect(dbp)
    db.execute('create table kernels (name text, start integer, end integer, duration integer, vgpr_count integer, accum_vgpr_count integer,'
               ' sgpr_count integer, lds_size integer, scratch_size integer, grid_x integer, workgroup_x integer, queue_id integer)')
    db.execute('create table regions (name text, category text, start integer, end integer, duration integer, extdata text)')
    t = 0
    for it in range(6):             # six iterations: anchor, a long kernel on queue 1, a short one on queue 2 overlapping it, a gap
        for name, dur, q, gap in (('light_rows_kernel(float const*)', 5000, 1, 0), ('env_trace_kernel<false>(TraceLaunch)', 2400000, 1, 1000),
                                  ('bvh_fit_kernel(float const*)', 160000, 2, -2300000), ('adam_step_kernel(AdamTable)', 30000, 1, 2200000)):
            t += gap
            db.execute('insert into kernels values (?,?,?,?,?,?,?,?,?,?,?,?)', (name, t, t + dur, dur, 32, 0, 64, 16384, 0, 524288, 256, q))
            t += dur
        db.execute('insert into regions values (?,?,?,?,?,?)', ('roctxThreadRangeA', 'MARKER_CORE_RANGE_API', t, t + 700000, 700000, '{"message":"nvdr_env_shade_fwd"}'))
        t += 3000000
    db.commit()
    db.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_summary.py'), dbp], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert '| void env_trace_kernel' not in out.stdout and '| env_trace_kernel<false>(TraceLaunch) | 6 |' in out.stdout
    assert '| nvdr_env_shade_fwd | 6 | 4200.0 | 700.00 |' in out.stdout
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_iteration.py'), dbp], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith('iteration of') and '4 dispatches' in lines[0] and 'queue_id' in lines[0]
    assert 'light_rows_kernel' in lines[2] and float(lines[2].split()[0]) == 0.0
    fit = [l for l in lines if 'bvh_fit_kernel' in l][0].split()
    assert fit[3] == '2' and float(fit[2]) == 0.0          # on the other queue, no idle time before it (the traversal is still running)


def test_broadcast_pixels_column_sum_and_composite_reference():
    """Harness helpers that run on any device: the stride-0 broadcast with its two-step column-sum backward, and the
    torch formulation of the fused composite (render.py:119-127) the HIP op is tested against."""
    from nvdiffrecmc_amd.trainer import _broadcast_pixels
    from nvdiffrecmc_amd.renderutils import torch_ref
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.rand(2, 16, 32, 3, generator=g, dtype=torch.float64)       # 1024 pixels: the k = 256 path
    (_broadcast_pixels.apply(x, 2, 16, 32) * w).sum().backward()
    assert torch.allclose(x.grad, w.sum((0, 1, 2)), rtol=1e-12)
    x.grad = None
    w = torch.rand(1, 5, 7, 3, generator=g, dtype=torch.float64)         # 35 pixels: the k = 1 path
    (_broadcast_pixels.apply(x, 1, 5, 7) * w).sum().backward()
    assert torch.allclose(x.grad, w.sum((0, 1, 2)), rtol=1e-12)
    d4, s4 = torch.rand(1, 4, 4, 4, generator=g) + 0.1, torch.rand(1, 4, 4, 4, generator=g) + 0.1
    kd, ks = torch.rand(1, 4, 4, 3, generator=g), torch.rand(1, 4, 4, 3, generator=g)
    ref = (d4[..., :3] / d4[..., 3:]) * kd * (1 - ks[..., 2:3]) + s4[..., :3] / s4[..., 3:]
    assert torch.allclose(torch_ref.shade_composite(d4, s4, kd, ks, 'pbr'), ref, rtol=1e-6)
    assert torch.allclose(torch_ref.shade_composite(d4[..., :3], s4[..., :3], kd, ks, 'diffuse'), d4[..., :3] * kd, rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# round 2

def _cpu_lbvh(v, t):
    """A tiny median-split binary tree in the record layout of nvdiffrecmc_amd/csrc/bvh.h (32-B nodes, 16-bit quantised
    child boxes, conservative rounding) -- only to exercise oracle_bvh2_walk without a GPU."""
    import numpy as np
    v, t = v.numpy().astype(np.float32), t.numpy()
    tri = v[t]                                                    # [T,3,3]
    lo, hi = v.min(0), v.max(0)
    scale = max(float((hi - lo).max()), float(np.abs(lo).max()), float(np.abs(hi).max()))
    pad = np.float32(1e-5 * scale)
    g_lo = (lo - 2 * pad).astype(np.float32)
    g_scale = (np.float32(65531.0) / np.maximum((hi + 2 * pad) - g_lo, 1e-6 * scale + 1e-30)).astype(np.float32)
    order = np.argsort(tri.mean(1)[:, 0], kind='stable')
    T = len(order)
    trirec = np.zeros((T, 12), np.float32)
    for k, o in enumerate(order):
        a, b, c = tri[o]
        trirec[k, 0:3], trirec[k, 3:6], trirec[k, 6:9] = a, b - a, c - a
    qbox = np.zeros((T, 6), np.int64)
    for k, o in enumerate(order):
        mn, mx = tri[o].min(0) - pad, tri[o].max(0) + pad
        qbox[k, :3] = np.clip(np.floor((mn - g_lo) * g_scale + 2.0) - 1, 0, 65535)
        qbox[k, 3:] = np.clip(np.ceil((mx - g_lo) * g_scale + 2.0) + 1, 0, 65535)
    nodes = []

    def build(a, b):                                              # leaves [a, b) -> (reference, box)
        if b - a == 1:
            return ~a, qbox[a]
        idx = len(nodes)
        nodes.append(None)
        m = (a + b) // 2
        cl, bl = build(a, m)
        cr, br = build(m, b)
        w = [int(bl[0]) | int(bl[1]) << 16, int(bl[2]) | int(bl[3]) << 16, int(bl[4]) | int(bl[5]) << 16,
             int(br[0]) | int(br[1]) << 16, int(br[2]) | int(br[3]) << 16, int(br[4]) | int(br[5]) << 16,
             cl & 0xffffffff, cr & 0xffffffff]
        nodes[idx] = w
        return idx, np.concatenate([np.minimum(bl[:3], br[:3]), np.maximum(bl[3:], br[3:])])

    build(0, T)
    return np.array(nodes, dtype=np.uint32), trirec, g_lo, g_scale


def test_cpu_walk_of_a_bvh2_equals_bruteforce():
    """oracle_bvh2_walk (the CPU checker of the GPU's traversal counters) answers like the brute-force loop."""
    from oracle import oracle as orc
    from nvdiffrecmc_amd import scene as sc
    mesh = sc.load_mesh('spot')
    v, t = mesh['v_pos'], mesh['t_pos_idx'][:600].contiguous()
    nodes, trirec, g_lo, g_scale = _cpu_lbvh(v, t)
    g = torch.Generator().manual_seed(4)
    ro = (torch.randn(20000, 3, generator=g) * 0.4).contiguous()
    rd = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1).contiguous()
    rd[:300] = torch.eye(3).repeat(100, 1)
    ref = orc.visibility(v, t, ro, rd, n_threads=2)
    vis, n_node, n_tri = orc.bvh2_walk(nodes, trirec, list(g_lo), list(g_scale), ro, rd, n_threads=2)
    assert torch.equal(vis, ref)
    assert 0.02 < ref.float().mean().item() < 0.999
    assert n_node > ro.shape[0] and n_tri > 0 and n_node < ro.shape[0] * 600


def test_stack_bound_formula():
    """nvdr_stack_bound (csrc/bvh.h) restated and compiled: h_max = 30 + ceil(log2 n) levels of the Karras tree + the NVDR_TREELET_CAP = 16
    a rebuilt treelet may add, capped at NVDR_STACK_MAX = 104 (one stack entry per level for the binary and the eight-wide walk alike)."""
    import math
    import subprocess
    import tempfile
    cases = ((1, 46), (2, 47), (10688, 60), (171008, 64), (684032, 66), (1 << 24, 70), ((1 << 28) - 1, 74))
    for n, expect in cases:
        h = 30 + (0 if n <= 1 else math.ceil(math.log2(n))) + 16
        assert min(h, 104) == expect
    # ... and what the header itself computes (host-side inline function, no GPU needed)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'b.cpp')
        open(src, 'w').write('#include <stdio.h>\n#include <stdint.h>\n#define NVDR_STACK_MAX 104\n#define NVDR_TREELET_CAP 16\n'
                             + _between(open(os.path.join(ROOT, 'nvdiffrecmc_amd', 'csrc', 'bvh.h')).read(), 'static inline int nvdr_stack_bound', '\n}\n') + '\n}\n'
                             + 'int main(){long long v[] = {%s}; for (auto x : v) printf("%%d ", nvdr_stack_bound(x)); return 0;}\n' % ', '.join('%dll' % n for n, _ in cases))
        exe = os.path.join(d, 'b')
        subprocess.run(['g++', '-std=c++17', src, '-o', exe], check=True)
        got = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [e for _, e in cases]


def test_live_list_segments_hold_what_their_wavefronts_can_append():
    """live_segment_capacity (csrc/env_shade.hip; host and device use the one formula): generation wavefront w takes the groups w, w + W, ...
    and appends to segment w % 16, so a segment must hold the slots of all groups of its wavefronts -- for every launch shape, also ragged ones."""
    import subprocess
    import tempfile
    text = open(os.path.join(ROOT, 'nvdiffrecmc_amd', 'csrc', 'env_shade.hip')).read()
    body = _between(text, '__host__ __device__ static inline unsigned long long live_segment_capacity', '\n}\n') + '\n}\n'
    body = body.replace('__host__ __device__ ', '')
    cases = [(10240, 2097152, 128), (10240, 454278, 128), (16, 1, 128), (40, 7, 2048), (4096, 4097, 128), (7, 1000, 512), (10240, 61635, 128), (12, 100000, 128)]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'c.cpp')
        open(src, 'w').write('#include <stdio.h>\n#define NVDR_LIVE_SEGS 16\n' + body +
                             'int main(){unsigned long long v[][3] = {%s}; for (auto &x : v) printf("%%llu ", live_segment_capacity(x[0], x[1], x[2])); return 0;}\n'
                             % ', '.join('{%dull, %dull, %dull}' % c for c in cases))
        exe = os.path.join(d, 'c')
        subprocess.run(['g++', '-std=c++17', src, '-o', exe], check=True)
        caps = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    for (waves, groups, slots), cap in zip(cases, caps):
        assert cap % 128 == 0
        # the most a segment can receive: every slot of every group of its wavefronts is a live ray
        per_wave = np.array([len(range(w, groups, waves)) for w in range(waves)], dtype=np.int64)
        per_seg = np.array([per_wave[s::16].sum() for s in range(16)]) * slots
        assert per_seg.max() <= cap, (waves, groups, slots, int(per_seg.max()), cap)
        assert cap == (-(-waves // 16) * -(-groups // waves) * slots + 127) // 128 * 128


def _between(text, start, end):
    i = text.index(start)
    return text[i:text.index(end, i)]


def test_render_layer_restatement_interpolates_like_nvdiffrast():
    """oracle/render_layer_ref.py: barycentric convention (u, v weigh vertex 0 and 1), zero background, interleaved derivative
    layout, and the (z/w, |dz|) pair of render.py:228-234 on a hand-made one-triangle rast."""
    from oracle import render_layer_ref as rl
    attr = torch.tensor([[1.0, 10.0], [2.0, 20.0], [4.0, 40.0]])
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    rast = torch.zeros(1, 1, 3, 4)
    rast[0, 0, 0] = torch.tensor([1.0, 0.0, 0.5, 1.0])        # all weight on vertex 0
    rast[0, 0, 1] = torch.tensor([0.25, 0.5, 0.5, 1.0])       # 0.25 a0 + 0.5 a1 + 0.25 a2
    db = torch.zeros(1, 1, 3, 4)
    db[0, 0, 1] = torch.tensor([0.1, 0.0, 0.0, 0.2])          # du/dX = 0.1, dv/dY = 0.2
    out, d = rl.interpolate(attr, rast, tri, db)
    assert torch.allclose(out[0, 0, 0], attr[0]) and torch.allclose(out[0, 0, 1], torch.tensor([2.25, 22.5]))
    assert torch.equal(out[0, 0, 2], torch.zeros(2)) and torch.equal(d[0, 0, 2], torch.zeros(4))
    # (dA0/dX, dA0/dY, dA1/dX, dA1/dY) = (0.1 (a0 - a2), 0.2 (a1 - a2)) per channel
    assert torch.allclose(d[0, 0, 1], torch.tensor([0.1 * (1 - 4), 0.2 * (2 - 4), 0.1 * (10 - 40), 0.2 * (20 - 40)]))


# ---------------------------------------------------------------------------------------------------------------------
# chunk dealing of the traversal kernel (csrc/trace_kernel.h ChunkDealer): a sequential model of claim() under random
# wave interleavings -- every chunk of the list must be handed out exactly once for every grid the launcher can start

def _deal(n_rays, n_waves, rng):
    QC, Q = 256, 64
    n_chunks = (n_rays + QC - 1) // QC
    counters = [0] * Q
    live = list(range(n_waves))
    claimed = []
    while live:
        w = live[rng.randrange(len(live))]
        sub = w % Q
        j = counters[sub]
        counters[sub] += 1
        c = j * Q + sub
        if c < n_chunks:
            claimed.append(c)
        else:
            live.remove(w)
    return n_chunks, claimed


def test_chunk_dealing_hands_out_every_chunk_exactly_once():
    import random
    rng = random.Random(7)
    for n_rays in (0, 1, 255, 256, 257, 4095, 4096, 4097, 70001, 2_000_003):
        blocks = max(1, min(2048, (n_rays + 255) // 256))            # the launcher's grid (env_shade.hip)
        n_chunks, claimed = _deal(n_rays, 4 * blocks, rng)
        assert sorted(claimed) == list(range(n_chunks)), n_rays
