"""LBVH build + traversal on the GPU vs the oracle's brute force: bit-exact visibility and closest hits."""
import math

import pytest
import torch

from oracle import oracle as orc
from nvdiffrecmc_amd import scene as sc
from tests.gpu_util import make_ctx

pytestmark = pytest.mark.gpu
NT = orc.max_threads()


def _rays(n, seed, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    ro = torch.randn(n, 3, generator=g) * scale
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    return ro.contiguous(), rd.contiguous()


@pytest.mark.parametrize('mesh_name', ['bob', 'spot'])
def test_visibility_bit_exact_vs_bruteforce(mesh_name, dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    ctx = make_ctx(mesh, dev)
    info = ctx.bvh_info()
    assert info['n_tris'] == mesh['t_pos_idx'].shape[0] and 1 <= info['height'] <= 64
    ro, rd = _rays(200000, 1)
    # add axis-parallel and surface-grazing rays (edge cases of the slab test)
    rd[:3000] = torch.eye(3).repeat(1000, 1)
    v = mesh['v_pos']
    ro[3000:6000] = v[torch.randint(0, v.shape[0], (3000,), generator=torch.Generator().manual_seed(2))]
    ref = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    got, cnt = ou.trace_visibility(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got.cpu(), ref), '%d of %d rays differ' % (int((got.cpu() != ref).sum()), ro.shape[0])
    assert 0.05 < ref.float().mean().item() < 0.95
    assert cnt[0].item() > 0 and cnt[1].item() > 0


def test_closest_hit_vs_bruteforce(dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    ctx = make_ctx(mesh, dev)
    mv, _, _ = sc.camera(2, 8)
    ro, rd = sc.primary_rays(mv, 96, 96)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    t, tri, uv = orc.closest(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    gt, gtri, guv = ou.trace_closest(ctx, ro.to(dev), rd.to(dev))
    gt, gtri, guv = gt.cpu(), gtri.cpu(), guv.cpu()
    assert torch.equal((gtri >= 0), (tri >= 0))
    hit = tri >= 0
    assert hit.float().mean().item() > 0.1
    assert torch.equal(gt[hit], t[hit])                       # same predicate, same arithmetic: identical distances
    same = gtri == tri                                        # ties between coplanar neighbours may pick either triangle
    assert same[hit].float().mean().item() > 0.999
    assert torch.equal(guv[hit & same], uv[hit & same])


def test_refit_equals_rebuild_and_single_triangle(dev):
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('spot')
    v, t = mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v, t, rebuild=1)
    g = torch.Generator().manual_seed(5)
    v2 = (mesh['v_pos'] * 1.1 + 0.01 * torch.randn(mesh['v_pos'].shape, generator=g)).contiguous()
    ou.optix_build_bvh(ctx, v2.to(dev), t, rebuild=0)          # refit the old topology to moved vertices
    ro, rd = _rays(50000, 3, 0.35)
    ref = orc.visibility(v2, mesh['t_pos_idx'], ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    ou.optix_build_bvh(ctx, v2.to(dev), t, rebuild=1)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    with pytest.raises(RuntimeError, match='refit'):
        ou.optix_build_bvh(ctx, v2.to(dev), t[:100].contiguous(), rebuild=0)
    # degenerate sizes: one triangle, two triangles
    for nt in (1, 2):
        tv = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
        tt = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)[:nt].contiguous()
        ou.optix_build_bvh(ctx, tv.to(dev), tt.to(dev), rebuild=1)
        ro2 = torch.tensor([[0.2, 0.2, 1.0], [0.2, 0.2, 1.0], [-1.0, 0.2, 0.2]])
        rd2 = torch.tensor([[0.0, 0, -1.0], [0.0, 0, 1.0], [1.0, 0, 0]])
        ref2 = orc.visibility(tv, tt, ro2, rd2)
        assert torch.equal(ou.trace_visibility(ctx, ro2.to(dev), rd2.to(dev)).cpu(), ref2)


def test_refit_of_a_moved_684k_mesh_answers_like_a_rebuild(dev):
    """The refit policy of the harness (trainer rebuild_every, round 6): the 684 032-triangle mesh with every vertex moved by a few
    learning-rate steps' worth of noise, eight refits in a row on the topology of ONE rebuild -- visibility through the production
    kernel after every refit == the binary walk == what a fresh rebuild answers, and on a ray sample == the oracle's brute force."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    v0, t = sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], 3)
    assert t.shape[0] == 684032
    ctx, fresh = ou.OptiXContext(), ou.OptiXContext()
    ou.optix_build_bvh(ctx, v0.to(dev), t.to(dev), rebuild=1)
    g = torch.Generator().manual_seed(17)
    ro, rd = _rays(300000, 5, 0.35)
    ro, rd = ro.to(dev), rd.to(dev)
    v = v0.clone()
    for it in range(8):
        v = (v + 2e-4 * torch.randn(v.shape, generator=g)).contiguous()           # ~20 Adam steps at the position rate of the benchmark
        ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=0)
        got = ou.trace_visibility_wide(ctx, ro, rd)
        assert torch.equal(got, ou.trace_visibility(ctx, ro, rd))
        if it in (0, 7):
            ou.optix_build_bvh(fresh, v.to(dev), t.to(dev), rebuild=1)
            assert torch.equal(got, ou.trace_visibility_wide(fresh, ro, rd)), 'refit %d differs from a rebuild' % it
    ref = orc.visibility(v, t, ro[:4000].cpu(), rd[:4000].cpu(), n_threads=NT)
    assert torch.equal(got[:4000].cpu(), ref)
    _check_oct_tree(ctx, v, t)
    ctx.check()


def test_build_rejects_empty_mesh_like_the_reference(dev):
    from nvdiffrecmc_amd import optixutils as ou
    ctx = ou.OptiXContext()
    with pytest.raises(AssertionError, match='empty training triangle mesh'):
        ou.optix_build_bvh(ctx, torch.zeros(3, 3, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev), 1)


def test_large_mesh_build_and_trace(dev):
    """DMTet-sized stand-in: bob subdivided twice (171k triangles)."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    v, t = sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], 2)
    assert t.shape[0] == 10688 * 16
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    ro, rd = _rays(20000, 9)
    ref = orc.visibility(v, t, ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)


# ---------------------------------------------------------------------------------------------------------------------
# round 2: independent predicate check, degenerate trees, canonical traversal counters

def _fp64_visibility(v, t, ro, rd, eps=1e-5, chunk=4096):
    """INDEPENDENT oracle of the hit predicate (SURVEY 8c deliverable 1): textbook Moeller-Trumbore in float64 torch ops,
    written against the mathematical definition only -- it shares no code with include/nvdr_raytri.h (which the HIP
    kernels, oracle/nvdr_oracle.c and the reference shim all include).  Returns (vis uint8 [R], clear bool [R]):
    `clear` rays are those for which EVERY triangle is either hit or missed by a margin > eps in all four quantities
    (barycentric u, v, 1-u-v and distance t, scaled by the size of the terms they are computed from), i.e. rays whose
    answer no correct implementation can disagree about."""
    dev = ro.device
    v = v.double()
    v0, v1, v2 = v[t[:, 0].long()], v[t[:, 1].long()], v[t[:, 2].long()]
    e1, e2 = (v1 - v0)[None], (v2 - v0)[None]                         # [1,T,3]
    le1, le2 = e1.norm(dim=-1), e2.norm(dim=-1)
    vis = torch.empty(ro.shape[0], dtype=torch.uint8, device=dev)
    clear = torch.empty(ro.shape[0], dtype=torch.bool, device=dev)
    for b in range(0, ro.shape[0], chunk):
        o, d = ro[b:b + chunk].double()[:, None, :], rd[b:b + chunk].double()[:, None, :]   # [r,1,3]
        pv = torch.cross(d.expand(-1, e2.shape[1], -1), e2.expand(d.shape[0], -1, -1), dim=-1)
        det = (e1 * pv).sum(-1)
        sgn = torch.where(det < 0, -1.0, 1.0)
        adet = det * sgn
        tv = o - v0[None]
        un = (tv * pv).sum(-1) * sgn
        qv = torch.cross(tv, e1.expand(tv.shape[0], -1, -1), dim=-1)
        vn = (d * qv).sum(-1) * sgn
        tn = (e2 * qv).sum(-1) * sgn
        wn = adet - un - vn
        hit = (un >= 0) & (vn >= 0) & (wn >= 0) & (tn > 0) & (adet > 0)
        # scales of the four numerators: |d||e1||e2| for det-like terms with |tv| for the ones containing the origin
        ld, lt = d.norm(dim=-1), tv.norm(dim=-1)
        s_b = eps * le1 * le2 * ld * torch.clamp(lt / torch.minimum(le1, le2).clamp(min=1e-30), min=1.0)
        s_t = eps * le1 * le2 * lt.clamp(min=1e-30)
        clear_hit = (un > s_b) & (vn > s_b) & (wn > s_b) & (tn > s_t)
        clear_miss = (un < -s_b) | (vn < -s_b) | (wn < -s_b) | (tn < -s_t)
        vis[b:b + chunk] = (~hit.any(dim=1)).to(torch.uint8)
        clear[b:b + chunk] = (clear_hit | clear_miss).all(dim=1)
    return vis, clear


@pytest.mark.parametrize('mesh_name', ['bob', 'spot'])
def test_predicate_vs_independent_fp64(mesh_name, dev):
    """>= 1 M random + grazing rays: the traversal's answer must equal the fp64 definition on every ray whose margins
    exceed 1e-5; the rest (rays through an edge / vertex / grazing a plane within rounding) are counted and reported."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    ctx = make_ctx(mesh, dev)
    v, t = mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev)
    n_rand, n_graze = 800000, 250000
    ro, rd = _rays(n_rand, 11, 0.4)
    g = torch.Generator().manual_seed(12)
    # grazing rays: from random origins THROUGH points on triangle edges and vertices (the predicate's decision boundary)
    ti = torch.randint(0, t.shape[0], (n_graze,), generator=g)
    tri = mesh['t_pos_idx'][ti].long()
    a, b = mesh['v_pos'][tri[:, 0]], mesh['v_pos'][tri[:, 1]]
    s = torch.rand(n_graze, 1, generator=g)
    s[: n_graze // 5] = 0.0                                            # exactly through a vertex
    target = a + s * (b - a)
    go = torch.randn(n_graze, 3, generator=g) * 0.8
    gd = torch.nn.functional.normalize(target - go, dim=-1)
    ro, rd = torch.cat([ro, go]).contiguous().to(dev), torch.cat([rd, gd]).contiguous().to(dev)
    got = ou.trace_visibility(ctx, ro, rd)
    assert torch.equal(ou.trace_visibility_wide(ctx, ro, rd), got)     # the production (wide-node) kernel answers identically
    ref, clear = _fp64_visibility(v, t, ro, rd)
    diff = got != ref
    n_clear = int(clear.sum())
    assert n_clear > 0.9 * n_rand                                      # the random rays are (almost) all unambiguous
    assert int((diff & clear).sum()) == 0, '%d clear rays disagree with the fp64 predicate' % int((diff & clear).sum())
    n_amb = int((~clear).sum())
    print('\n[%s] %d rays: %d clear (0 disagree), %d within 1e-5 of a decision boundary, of which %d (%.2f%%) disagree with fp64'
          % (mesh_name, ro.shape[0], n_clear, n_amb, int((diff & ~clear).sum()), 100.0 * int((diff & ~clear).sum()) / max(n_amb, 1)))
    assert n_amb > 1000                                                # the grazing set really probes the boundary
    assert 0.05 < ref.float().mean().item() < 0.95
    ctx.check()


def _degenerate_fan(n, seed):
    """n triangles that all share ONE centroid (identical Morton keys: the tree is built from the index tie-break alone)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, 3, generator=g) * 0.3
    b = torch.randn(n, 3, generator=g) * 0.3
    v = torch.stack([a, b, -(a + b)], dim=1).reshape(-1, 3).contiguous()
    t = torch.arange(3 * n, dtype=torch.int32).view(n, 3).contiguous()
    return v, t


def _sliver_chain(n):
    """n needle triangles strung along a line: long, thin, overlapping boxes, keys that differ in one axis only."""
    i = torch.arange(n, dtype=torch.float32)
    x0 = i / n * 2.0 - 1.0
    a = torch.stack([x0, torch.zeros(n), torch.zeros(n)], -1)
    b = torch.stack([x0 + 0.5, 1e-4 * torch.ones(n), torch.zeros(n)], -1)
    c = torch.stack([x0 + 0.5, torch.zeros(n), 1e-4 * torch.cos(i)], -1)
    v = torch.stack([a, b, c], dim=1).reshape(-1, 3).contiguous()
    t = torch.arange(3 * n, dtype=torch.int32).view(n, 3).contiguous()
    return v, t


@pytest.mark.parametrize('kind', ['fan', 'slivers'])
def test_degenerate_meshes_match_bruteforce(kind, dev):
    """>= 100 k triangles on one centroid / a long sliver chain: deep tie-break chains must stay inside the proven stack
    bound (csrc/bvh.h) -- visibility equals brute force and the context reports no overflow."""
    from nvdiffrecmc_amd import optixutils as ou
    v, t = _degenerate_fan(100000, 3) if kind == 'fan' else _sliver_chain(120000)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    info = ctx.bvh_info()
    # h <= 30 + ceil(log2 n) for the Karras tree (+ the 16 levels a rebuilt treelet may add, csrc/bvh.h); the stacks hold the bound
    assert info['height'] <= 30 + 17 + 16 and info['stack_max'] >= info['height'] and info['stack_max'] == 30 + 17 + 16
    # half random rays (near and far origins), half aimed at interior points of random triangles (certain hits, deep descents)
    ro, rd = _rays(3000, 21, 0.3)
    ro2, rd2 = _rays(3000, 22, 1.5)
    g = torch.Generator().manual_seed(23)
    ti = torch.randint(0, t.shape[0], (3000,), generator=g)
    tri = v[t[ti].long()]                                            # [R,3,3]
    w = torch.rand(3000, 3, generator=g) + 0.05
    w = w / w.sum(-1, keepdim=True)
    target = (tri * w[..., None]).sum(1)
    ro3 = target + torch.nn.functional.normalize(torch.randn(3000, 3, generator=g), dim=-1) * 2.0
    rd3 = torch.nn.functional.normalize(target - ro3, dim=-1)
    ro, rd = torch.cat([ro, ro2, ro3]).contiguous(), torch.cat([rd, rd2, rd3]).contiguous()
    ref = orc.visibility(v, t, ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    assert torch.equal(ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)    # the production wide walk
    ctx.check()
    assert 0.0 < ref.float().mean().item() < 1.0
    assert ref[6000:].float().mean().item() < 0.5                    # the aimed rays do hit


def test_stack_overflow_is_reported_not_silent(dev, monkeypatch):
    """NVDR_DEBUG bit 32 pretends the stack holds 13 entries: the walk of a real mesh overflows it, and the library must
    say so on the next call instead of returning a wrong visibility quietly."""
    from nvdiffrecmc_amd import optixutils as ou
    monkeypatch.setenv('NVDR_DEBUG', '32')
    ctx = ou.OptiXContext()
    monkeypatch.delenv('NVDR_DEBUG')
    v, t = _degenerate_fan(50000, 5)
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    assert ctx.bvh_info()['stack_max'] == 13
    ro, rd = _rays(20000, 22, 0.05)
    ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev))
    with pytest.raises(RuntimeError, match='overflow'):
        ctx.check()
    with pytest.raises(RuntimeError, match='overflow'):
        ou.trace_visibility(ctx, ro.to(dev), rd.to(dev))


def test_canonical_counters_equal_cpu_walk_of_exported_tree(dev):
    """the node-visit / triangle-test counts behind the roofline figure must equal a CPU traversal of the SAME
    tree.  The GPU counting kernel (binary any-hit walk) and oracle_bvh2_walk over nvdr_bvh_export agree exactly."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    ctx = make_ctx(mesh, dev)
    info = ctx.bvh_info()
    nodes, trirec = ctx.bvh_export()
    ro, rd = _rays(100000, 31, 0.35)
    got, cnt = ou.trace_visibility(ctx, ro.to(dev), rd.to(dev), count=True)
    vis, n_node, n_tri = orc.bvh2_walk(nodes, trirec, info['grid_lo'], info['grid_scale'], ro, rd, n_threads=NT)
    assert torch.equal(got.cpu(), vis)
    assert int(cnt[0]) == 2 * n_node and int(cnt[1]) == n_tri          # the kernel counts two box tests per node visit
    assert 5 < n_node / ro.shape[0] < 60


def test_production_kernel_equals_binary_walk_at_odd_list_lengths(dev):
    """The production shadow-ray kernel claims chunks of 256 rays from 64 counters (csrc/trace_kernel.h): it must answer like
    the binary walk, bit for bit, also when the list is shorter than the counters serve and when it does not divide by the
    chunk size."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    ctx = make_ctx(mesh, dev)
    for n in (1, 63, 257, 2048 + 17, 4096, 5000, 300001):
        ro, rd = _rays(n, 31 + n)
        ro, rd = ro.to(dev), rd.to(dev)
        ref = ou.trace_visibility(ctx, ro, rd)
        got = ou.trace_visibility_wide(ctx, ro, rd)
        assert torch.equal(got, ref), '%d rays: %d differ' % (n, int((got != ref).sum()))
    ctx.check()


def test_axis_parallel_rays_cost_what_other_rays_cost(dev):
    """A ray that does not move along an axis has an infinite inverse direction there; round 2 found that the fused slab test then
    stopped culling on that axis (NaN planes), so an EXACTLY axis-parallel ray -- the light sample at the pole of the probe is
    (0, 1, -0) once in ~1e8 samples -- walked a large part of the tree: hundreds of thousands of steps for one lane on a big mesh
    (the "slow mode" of rounds 1-2).  make_grid_ray caps |1/d|: such rays must answer like brute force AND visit about as many
    boxes as other rays (counted by the binary walk: this is a deterministic check, not a timing)."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh('bob')
    ctx = make_ctx(mesh, dev)
    g = torch.Generator().manual_seed(11)
    v = mesh['v_pos']
    n = 6000
    ro = (v[torch.randint(0, v.shape[0], (n,), generator=g)] * 0.999).contiguous()      # just inside the surface: long walks
    dirs = torch.tensor([[0.0, 1.0, -0.0], [-0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [-1.0, -0.0, 0.0], [0.0, -0.0, 1.0], [-0.0, 0.0, -1.0]])
    rd = dirs.repeat(n // 6, 1).contiguous()
    ref = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    got, cnt = ou.trace_visibility(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got.cpu(), ref), '%d of %d axis-parallel rays differ' % (int((got.cpu() != ref).sum()), n)
    assert torch.equal(ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    ro2, rd2 = _rays(n, 12)
    _, cnt2 = ou.trace_visibility(ctx, ro2.to(dev), rd2.to(dev), count=True)
    per_ray, per_ray2 = cnt[0].item() / n, cnt2[0].item() / n
    assert per_ray < 4.0 * per_ray2 + 50.0, 'axis-parallel rays test %.0f boxes each, random rays %.0f' % (per_ray, per_ray2)


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the large-mesh preset and the production walk's own cost counters

def test_684k_triangle_mesh_real_shadow_rays_vs_bruteforce(dev):
    """bench.py --config dmtet800 (bob subdivided three times, 684 032 triangles -- the tree does not fit the L2s): >= 50 k shadow
    rays of a REAL env-shade launch (the samples the raygen program draws for ~400 pixels of the 800x800 view, from the oracle's
    sample generator: bit-identical directions) plus exact pole rays (0, 1, -0) through the PRODUCTION kernel and the binary
    walk == the oracle's brute force over all 684 k triangles; then the same pixels through the whole env-shade launch."""
    from nvdiffrecmc_amd import optixutils as ou
    from tests.test_gpu_fullsize import _gpu_scene, _shade
    res, n, seed = 800, 8, 4
    S = n * n
    mesh, ctx, kw, perms = _gpu_scene('bob', res, n, dev, view=5, subdiv=3)
    assert ctx.bvh_info()['n_tris'] == 684032
    sub = torch.zeros_like(kw['mask'])
    sub[:, 9::18, 4::20] = kw['mask'][:, 9::18, 4::20]
    kws = dict(kw, mask=sub)
    cpu = {k: v.detach().cpu().contiguous() for k, v in kws.items()}
    P = res * res
    ones = torch.ones(P, 2 * S, dtype=torch.uint8)
    smp = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT, vis_in=ones, want_dbg=True)
    pix = (cpu['mask'].view(-1) > 0).nonzero().view(-1)
    assert 380 < pix.numel() < 480 and smp['covered'] == pix.numel()
    rd = smp['dbg'][pix][:, :, 0:3].reshape(-1, 3).contiguous()                      # [pixels * 2S, 3] in the oracle's sample order
    ro = cpu['ro'].view(-1, 3)[pix][:, None, :].expand(-1, 2 * S, -1).reshape(-1, 3).contiguous()
    assert rd.shape[0] >= 50000
    # exact pole rays: the light sample of stratum row 0 with a uniform draw of 0 (round 2's stragglers), one per 256 rays
    pole = torch.tensor([[0.0, 1.0, -0.0], [0.0, -1.0, 0.0]])
    rd[::256] = pole[0]
    rd[128::256] = pole[1]
    ref = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    got_w, (n_box, n_tri, n_ray, n_step) = ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got_w.cpu(), ref), '%d of %d rays differ (production kernel)' % (int((got_w.cpu() != ref).sum()), ref.numel())
    assert torch.equal(ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    assert n_ray == ref.numel() and 0.02 < 1.0 - ref.float().mean().item() < 0.9
    ctx.check()
    # the launch itself on these pixels: the oracle with the brute-force visibility of the un-poled rays
    rd0 = smp['dbg'][pix][:, :, 0:3].reshape(-1, 3).contiguous()
    vis_pix = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd0, n_threads=NT).view(-1, 2 * S)
    vis = ones.clone()
    vis[pix] = vis_pix
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT, vis_in=vis)
    d, s = _shade(ctx, kws, n, seed)
    from tests.util import assert_close
    assert_close(d, f['diff'], 2e-6, what='diff')
    assert_close(s, f['spec'], 2e-6, what='spec')
    ctx.check()


def test_production_walk_cost_of_axis_parallel_rays(dev):
    """Round 2's straggler rays, counted through the PRODUCTION walk (the counting build of env_trace_kernel, not the binary
    walk): a ray list salted with exactly axis-parallel rays must cost about what the same list costs without them, and an
    axis-parallel ray about what any other ray costs -- on bob and on a 171 k-triangle mesh."""
    from nvdiffrecmc_amd import optixutils as ou
    for subdiv in (0, 2):
        mesh = sc.load_mesh('bob')
        v, t = (mesh['v_pos'], mesh['t_pos_idx']) if subdiv == 0 else sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], subdiv)
        ctx = ou.OptiXContext()
        ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
        g = torch.Generator().manual_seed(11)
        n = 60000
        ro = (v[torch.randint(0, v.shape[0], (n,), generator=g)] * 0.999).contiguous()     # just inside the surface: long walks
        rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).contiguous()
        dirs = torch.tensor([[0.0, 1.0, -0.0], [-0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [-1.0, -0.0, 0.0], [0.0, -0.0, 1.0], [-0.0, 0.0, -1.0]])
        rd_ax = rd.clone()
        rd_ax[::10] = dirs.repeat(n // 60, 1)                                              # every 10th ray exactly axis-parallel
        vis0, c0 = ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev), count=True)
        vis1, c1 = ou.trace_visibility_wide(ctx, ro.to(dev), rd_ax.to(dev), count=True)
        assert torch.equal(vis0, ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)))
        assert torch.equal(vis1, ou.trace_visibility(ctx, ro.to(dev), rd_ax.to(dev)))
        only_ax = rd_ax[::10].contiguous()
        _, c2 = ou.trace_visibility_wide(ctx, ro[::10].contiguous().to(dev), only_ax.to(dev), count=True)
        per0, per1, per2 = c0[0] / n, c1[0] / n, c2[0] / (n // 10)
        print('\n[subdiv %d, %d triangles] production walk box tests per ray: random %.1f, salted %.1f, axis-parallel only %.1f; node steps %.2f'
              % (subdiv, t.shape[0], per0, per1, per2, c0[3] / n))
        assert c0[2] == n and c1[2] == n
        assert per1 < 2.0 * per0 and per2 < 2.0 * per0, (per0, per1, per2)
        ctx.check()


def _check_oct_tree(ctx, v, t):
    """Structural invariants of the eight-wide tree (csrc/bvh.h "oct"), checked on the host copy."""
    import numpy as np
    info = ctx.bvh_info()
    oct, tris8, cnt = ctx.bvh_export_oct()
    n = info['n_tris']
    assert cnt['triangles_placed'] == n and cnt['nodes_finished'] == cnt['nodes'] >= 1
    # every triangle exactly once (the original index travels in the record)
    orig = tris8[:, 9].view(np.int32)
    assert np.array_equal(np.sort(orig), np.arange(n))
    w0 = oct[:, 0:4]
    org = np.stack([w0[:, 0] & 0xffff, w0[:, 0] >> 16, w0[:, 1] & 0xffff], -1).astype(np.int64)
    ex = np.stack([(w0[:, 1] >> 16) & 15, (w0[:, 1] >> 20) & 15, (w0[:, 1] >> 24) & 15], -1).astype(np.int64)
    cbase, n_int = (w0[:, 2] & 0xfffffff).astype(np.int64), (w0[:, 2] >> 28).astype(np.int64)
    tbase, n_leaf = (w0[:, 3] & 0xfffffff).astype(np.int64), (w0[:, 3] >> 28).astype(np.int64)
    assert ((n_int + n_leaf >= 1) & (n_int + n_leaf <= 8)).all()

    def tiles(base, cnt_, lo, hi):              # the non-empty ranges, sorted, tile [lo, hi)
        m = cnt_ > 0
        o = np.argsort(base[m])
        b, c = base[m][o], cnt_[m][o]
        return b[0] == lo and (b[1:] == b[:-1] + c[:-1]).all() and b[-1] + c[-1] == hi
    assert cnt['nodes'] == 1 or tiles(cbase, n_int, 1, cnt['nodes'])
    assert tiles(tbase, n_leaf, 0, n)
    planes = oct[:, 4:16].copy().view(np.uint8).reshape(-1, 6, 8).astype(np.int64)   # lo.x, lo.y, lo.z, hi.x, hi.y, hi.z per slot
    lo = org[:, :, None] + (planes[:, 0:3, :] << ex[:, :, None])                   # [nodes, axis, slot] on the 16-bit grid
    hi = org[:, :, None] + (planes[:, 3:6, :] << ex[:, :, None])
    # leaf slots contain their triangle (exact vertices mapped to the grid)
    gl, gs = np.array(info['grid_lo']), np.array(info['grid_scale'])
    v0, e1, e2 = tris8[:, 0:3].astype(np.float64), tris8[:, 3:6].astype(np.float64), tris8[:, 6:9].astype(np.float64)
    pts = np.stack([v0, v0 + e1, v0 + e2], 1)
    g = (pts - gl) * gs + 2.0
    tmin, tmax = g.min(1), g.max(1)
    for node in np.nonzero(n_leaf > 0)[0]:
        for j in range(n_leaf[node]):
            k, tri = n_int[node] + j, tbase[node] + j
            assert (lo[node, :, k] <= tmin[tri] + 1e-3).all() and (hi[node, :, k] >= tmax[tri] - 1e-3).all(), (node, j)
    # an internal slot contains every TRIANGLE below the child node it points to.  Note that it need not contain the child's
    # own 8-bit slots: those are rounded outward in the child's frame, which may be coarser than what the parent's slot shows.
    # (The layout follows the Karras numbering of the binary tree, round 4: a child block may lie before or behind its parent, so the
    # nodes are visited in an order found by walking the tree from the root -- which also shows that every node is reachable once.)
    big = 1 << 30
    sub_lo = np.full((cnt['nodes'], 3), float(big))
    sub_hi = np.full((cnt['nodes'], 3), -float(big))
    order, seen, stack = [], np.zeros(cnt['nodes'], dtype=bool), [0]
    while stack:
        node = stack.pop()
        assert not seen[node]
        seen[node] = True
        order.append(node)
        stack.extend(range(cbase[node], cbase[node] + n_int[node]))
    assert seen.all()
    for node in reversed(order):                    # children before their parents
        for j in range(n_leaf[node]):
            tri = tbase[node] + j
            sub_lo[node] = np.minimum(sub_lo[node], tmin[tri])
            sub_hi[node] = np.maximum(sub_hi[node], tmax[tri])
        for k in range(n_int[node]):
            c = cbase[node] + k
            assert (lo[node, :, k] <= sub_lo[c] + 1e-3).all() and (hi[node, :, k] >= sub_hi[c] - 1e-3).all(), (node, k, c)
            sub_lo[node] = np.minimum(sub_lo[node], sub_lo[c])
            sub_hi[node] = np.maximum(sub_hi[node], sub_hi[c])
    return cnt['nodes'], float((n_int + n_leaf).mean())


@pytest.mark.parametrize('kind', ['bob', 'spot', 'single', 'pair', 'fan', 'subdiv1', 'subdiv2', 'dmtet64_init', 'dmtet64_mid'])
def test_oct_tree_invariants(kind, dev):
    """The eight-wide tree (budgets -> counts -> prefix sum -> emit, csrc/bvh.hip): every triangle placed once, every node reachable
    once from the root, children and leaves stored contiguously, every 8-bit box contains what it stands for -- on regular meshes,
    degenerate ones, and after a refit; the slot choice is the SAH-optimal one of the fit kernel's dynamic programme; the layout is a
    function of the tree alone (two builds give the same bytes)."""
    import numpy as np
    from nvdiffrecmc_amd import optixutils as ou
    if kind in ('bob', 'spot', 'dmtet64_init', 'dmtet64_mid'):
        m = sc.load_mesh(kind)
        v, t = m['v_pos'], m['t_pos_idx']
    elif kind in ('subdiv1', 'subdiv2'):
        m = sc.load_mesh('bob')
        v, t = sc.subdivide(m['v_pos'], m['t_pos_idx'], int(kind[-1]))
    elif kind == 'fan':
        v, t = _degenerate_fan(20000, 3)
    else:
        v = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
        t = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)[:1 if kind == 'single' else 2].contiguous()
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    nodes, fill = _check_oct_tree(ctx, v, t)
    print('\n[%s] %d triangles -> %d oct nodes, %.2f slots used per node' % (kind, t.shape[0], nodes, fill))
    if t.shape[0] > 8:
        assert nodes < t.shape[0] / 4 and fill > 4.5           # the optimal collapse: ~n / 4.9 nodes (the greedy rule of round 3: n / 3.3)
    if kind in ('bob', 'spot', 'fan', 'dmtet64_mid'):
        # deterministic layout: a second build (another context) gives the same bytes, and the walk through it answers like the binary walk
        oct_a, tris_a, _ = ctx.bvh_export_oct()
        ctx_b = ou.OptiXContext()
        ou.optix_build_bvh(ctx_b, v.to(dev), t.to(dev), rebuild=1)
        oct_b, tris_b, _ = ctx_b.bvh_export_oct()
        assert np.array_equal(oct_a, oct_b) and np.array_equal(tris_a, tris_b)
        g0 = torch.Generator().manual_seed(3)
        ro = (torch.rand(20000, 3, generator=g0) * 2 - 1).to(dev) * float(v.abs().max())
        rd = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g0), dim=-1).to(dev)
        assert torch.equal(ou.ops.trace_visibility_wide(ctx, ro, rd), ou.ops.trace_visibility(ctx, ro, rd))
    g = torch.Generator().manual_seed(1)
    v2 = (v * 1.05 + 0.01 * torch.randn(v.shape, generator=g)).contiguous()
    ou.optix_build_bvh(ctx, v2.to(dev), t.to(dev), rebuild=0)                   # refit: the oct tree is rebuilt over the new boxes
    _check_oct_tree(ctx, v2, t)
    ctx.check()


# ---------------------------------------------------------------------------------------------------------------------
# round 6: marching-tets extractions (tools/make_dmtet_mesh.py) -- what DMTet really hands optix_build_bvh (geometry/dmtet.py:202):
# irregular triangles, slivers where the surface grazes a grid vertex, floaters and internal sheets; not the regular patches of a
# subdivided mesh

@pytest.mark.parametrize('mesh_name', ['dmtet64_init', 'dmtet64_mid'])
def test_dmtet_extraction_visibility_vs_bruteforce(mesh_name, dev):
    """Random rays through the volume, axis-parallel rays, rays starting ON vertices, and the REAL shadow rays of an env-shade launch
    (the oracle's sample generator on a sparse pixel subset of the 800x800 view) through the production kernel and the binary walk ==
    the oracle's brute force over every triangle, bit for bit."""
    from nvdiffrecmc_amd import optixutils as ou
    from tests.test_gpu_fullsize import _gpu_scene
    res, n, seed = 800, 8, 6
    S = n * n
    mesh, ctx, kw, perms = _gpu_scene(mesh_name, res, n, dev, view=3)
    T = mesh['t_pos_idx'].shape[0]
    assert ctx.bvh_info()['n_tris'] == T and T > 70000
    ro, rd = _rays(60000, 11, scale=0.7)
    rd[:3000] = torch.eye(3).repeat(1000, 1)
    v = mesh['v_pos']
    ro[3000:6000] = v[torch.randint(0, v.shape[0], (3000,), generator=torch.Generator().manual_seed(2))]
    sub = torch.zeros_like(kw['mask'])
    sub[:, 7::29, 3::31] = kw['mask'][:, 7::29, 3::31]
    cpu = {k: v_.detach().cpu().contiguous() for k, v_ in dict(kw, mask=sub).items()}
    ones = torch.ones(res * res, 2 * S, dtype=torch.uint8)
    smp = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **cpu, perms=perms, n_samples_x=n, rnd_seed=seed, n_threads=NT, vis_in=ones, want_dbg=True)
    pix = (cpu['mask'].view(-1) > 0).nonzero().view(-1)
    assert pix.numel() > 100 and smp['covered'] == pix.numel()
    rd_s = smp['dbg'][pix][:, :, 0:3].reshape(-1, 3).contiguous()
    ro_s = cpu['ro'].view(-1, 3)[pix][:, None, :].expand(-1, 2 * S, -1).reshape(-1, 3).contiguous()
    ro, rd = torch.cat((ro, ro_s)).contiguous(), torch.cat((rd, rd_s)).contiguous()
    ref = orc.visibility(mesh['v_pos'], mesh['t_pos_idx'], ro, rd, n_threads=NT)
    got_w, (n_box, n_tri, n_ray, n_step) = ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got_w.cpu(), ref), '%d of %d rays differ (production kernel)' % (int((got_w.cpu() != ref).sum()), ref.numel())
    assert torch.equal(ou.trace_visibility(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    assert n_ray == ref.numel() and 0.02 < ref.float().mean().item() < 0.98
    print('\n[%s] %d triangles, %d rays: %.1f node steps, %.1f box tests, %.1f triangle tests per ray, %.0f %% unoccluded'
          % (mesh_name, T, ref.numel(), n_step / n_ray, n_box / n_ray, n_tri / n_ray, 100.0 * ref.float().mean().item()))
    ctx.check()


@pytest.mark.parametrize('mesh_name', ['bob', 'dmtet64_init'])
def test_split_walk_build_of_the_traversal_kernel_vs_bruteforce(mesh_name, dev, monkeypatch):
    """The traversal build with SPLIT WALKS in the drain (trace_kernel.h; the launcher starts it for launches of few rays per wavefront, by the
    context's last launch): forced on for every launch, its visibility == the plain build's == the oracle's brute force, on random rays and on
    a ray set made of a few thousand LONG walks (rays grazing the mesh: what a drain consists of) among short ones."""
    from nvdiffrecmc_amd import optixutils as ou
    mesh = sc.load_mesh(mesh_name)
    v, t = mesh['v_pos'], mesh['t_pos_idx']
    monkeypatch.setenv('NVDR_TUNING', '1')
    monkeypatch.setenv('NVDR_TRACE_SPLIT_MODE', '1')
    ctx_split = make_ctx(mesh, dev)
    monkeypatch.setenv('NVDR_TRACE_SPLIT_MODE', '0')
    ctx_plain = make_ctx(mesh, dev)
    monkeypatch.delenv('NVDR_TRACE_SPLIT_MODE')
    g = torch.Generator().manual_seed(23)
    ro, rd = _rays(90000, 23, scale=0.5)
    # grazing rays: from a vertex, along an edge of one of its triangles, lifted off the surface by a hair
    tri = torch.randint(0, t.shape[0], (20000,), generator=g)
    a, b = v[t[tri, 0].long()], v[t[tri, 1].long()]
    ro[:20000] = a + 1e-4 * torch.randn(20000, 3, generator=g)
    rd[:20000] = torch.nn.functional.normalize(b - a + 1e-3 * torch.randn(20000, 3, generator=g), dim=-1)
    ref = orc.visibility(v, t, ro, rd, n_threads=NT)
    for n in (ro.shape[0], 5000, 777):          # a launch of many rays per wavefront, of a few, of less than a wavefront each
        got_s = ou.trace_visibility_wide(ctx_split, ro[:n].to(dev), rd[:n].to(dev)).cpu()
        got_p = ou.trace_visibility_wide(ctx_plain, ro[:n].to(dev), rd[:n].to(dev)).cpu()
        assert torch.equal(got_s, ref[:n]), '%d of %d rays differ (split build)' % (int((got_s != ref[:n]).sum()), n)
        assert torch.equal(got_p, ref[:n])
    ctx_split.check()
    ctx_plain.check()


@pytest.mark.parametrize('top_nodes', [0, 1, 9, 64, 300])
def test_treetop_table_in_lds_leaves_the_visibility_unchanged(top_nodes, dev, monkeypatch):
    """The first nodes of the eight-wide tree in breadth-first order live in LDS in the shadow-ray kernel (bvh.h NVDR_TRACE_TOP_NODES, bvh.hip
    bvh_oct_top_kernel; a tuning switch, off by default): whatever the table's size -- none, the root alone, a level cut in the middle, more than the
    default -- the production kernel's visibility == the oracle's brute force, after a build and after a refit of moved vertices; a one-triangle
    mesh (a root without children) and the marching-tets mesh answer like the binary walk.  With a table, the counting build reports the node
    steps it served."""
    from nvdiffrecmc_amd import optixutils as ou
    from nvdiffrecmc_amd import _lib
    monkeypatch.setenv('NVDR_TUNING', '1')
    monkeypatch.setenv('NVDR_TRACE_TOP_NODES', str(top_nodes))
    mesh = sc.load_mesh('bob')
    v, t = mesh['v_pos'], mesh['t_pos_idx']
    ctx = make_ctx(mesh, dev)
    ro, rd = _rays(60000, 31, scale=0.5)
    ref = orc.visibility(v, t, ro, rd, n_threads=NT)
    got, (n_box, n_tri, n_ray, n_step) = ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev), count=True)
    assert torch.equal(got.cpu(), ref)
    assert torch.equal(ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev)).cpu(), ref)
    n_top = int(ou.trace_visibility_wide.last_counters[_lib.COUNTERS_BVH2 + 6])
    print('\ntreetop of %d entries: %.1f %% of %d node steps' % (top_nodes, 100.0 * n_top / max(n_step, 1), n_step))
    assert (n_top == 0) == (top_nodes == 0) and n_top <= n_step
    if top_nodes:
        assert n_top >= n_ray                       # every ray starts at the root, which is entry 0
    # moved vertices, refit: the table is rewritten behind the re-emitted nodes
    g = torch.Generator().manual_seed(5)
    v2 = v + 0.02 * torch.randn(v.shape, generator=g)
    ou.optix_build_bvh(ctx, v2.to(dev), t.to(dev).int(), rebuild=0)
    ref2 = orc.visibility(v2, t, ro, rd, n_threads=NT)
    assert torch.equal(ou.trace_visibility_wide(ctx, ro.to(dev), rd.to(dev)).cpu(), ref2)
    ctx.check()
    for other in ('single', 'dmtet64_init'):
        if other == 'single':
            vo = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
            to = torch.tensor([[0, 1, 2]], dtype=torch.int32)
        else:
            m = sc.load_mesh(other)
            vo, to = m['v_pos'], m['t_pos_idx']
        c2 = ou.OptiXContext()
        ou.optix_build_bvh(c2, vo.to(dev), to.to(dev).int(), rebuild=1)
        r_o, r_d = _rays(30000, 37, scale=0.6)
        assert torch.equal(ou.trace_visibility_wide(c2, r_o.to(dev), r_d.to(dev)), ou.ops.trace_visibility(c2, r_o.to(dev), r_d.to(dev))), other
        c2.check()
