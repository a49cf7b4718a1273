"""The literal CPU baseline of BASELINE.json configs[0] -- TEST / BENCH INFRASTRUCTURE, never on the product path.

"bob.json 128x128, 4 spp, brute-force PyTorch ray-triangle shadow test on CPU": the shadow rays of the direct-lighting pass
(kernel.cu:101-118: any hit in (0, 1e16), no culling) answered by plain torch ops -- a chunked [rays x triangles] Moeller-Trumbore
test under torch.set_num_threads(cores) -- around the restated raygen program (oracle/nvdr_oracle.c), which supplies the sample
directions and turns the visibility into radiance and gradients:

    pass A   oracle raygen with all-visible rays, recording every sample's direction            (oracle_env_shade, dbg)
    torch    any-hit visibility of the 2 S rays of every covered pixel by brute force            (this file)
    pass B   oracle forward + backward with that visibility                                      (oracle_env_shade, vis_in)

The textbook predicate in torch float32 (no fused multiply-adds, divisions) is NOT the bit-defined predicate of
include/nvdr_raytri.h: the two agree wherever a ray's decision margins are not within rounding of a triangle's edge / plane, which
tests/test_oracle_pins.py checks ("clear" rays), and `margin` below quantifies.
"""
import time

import torch


def shadow_rays_bruteforce_torch(v_pos, tris, ro, rd, chunk=256, want_margin=False):
    """uint8 [R]: 1 where the ray (ro, rd) hits NO triangle for t in (0, 1e16) (two-sided, no culling).  With want_margin also a
    float [R]: the smallest distance of any (ray, triangle) pair from a decision boundary of the test, in barycentric units."""
    t = tris.long()
    v0, v1, v2 = v_pos[t[:, 0]], v_pos[t[:, 1]], v_pos[t[:, 2]]
    e1, e2 = (v1 - v0)[None], (v2 - v0)[None]                       # [1,T,3]
    R = ro.shape[0]
    vis = torch.empty(R, dtype=torch.uint8)
    margin = torch.empty(R) if want_margin else None
    for b in range(0, R, chunk):
        o, d = ro[b:b + chunk, None, :], rd[b:b + chunk, None, :]   # [r,1,3]
        p = torch.cross(d.expand(-1, e2.shape[1], -1), e2.expand(d.shape[0], -1, -1), dim=-1)
        det = (e1 * p).sum(-1)
        ok = det.abs() > 1e-30
        inv = 1.0 / torch.where(ok, det, torch.ones_like(det))
        s = o - v0[None]
        u = (s * p).sum(-1) * inv
        q = torch.cross(s, e1.expand(s.shape[0], -1, -1), dim=-1)
        v = (d * q).sum(-1) * inv
        tt = (e2 * q).sum(-1) * inv
        hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (tt > 0) & (tt < 1e16)
        vis[b:b + chunk] = (~hit.any(dim=1)).to(torch.uint8)
        if want_margin:
            # a pair can only change its answer near u = 0, v = 0, u + v = 1 or t = 0, and only if the other conditions (loosened)
            # hold; pairs the ray is (nearly) parallel to are counted as undecided
            near = (u > -1e-3) & (v > -1e-3) & (u + v < 1 + 1e-3) & (tt > -1e-3)
            m = torch.minimum(torch.minimum(u.abs(), v.abs()), torch.minimum((1 - u - v).abs(), tt.abs()))
            m = torch.where(near & ok, m, torch.full_like(m, 1e9))
            m = torch.where(ok | ~near, m, torch.zeros_like(m))
            margin[b:b + chunk] = m.min(dim=1)[0]
    return (vis, margin) if want_margin else vis


def direct_lighting_torch_shadow(mesh, kw, n_samples_x, rnd_seed=0, diff_grad=None, spec_grad=None, n_threads=None, want=False):
    """BASELINE configs[0] end to end: returns (forward dict, backward dict or None, timing dict).  kw: oracle.scene_cpu.shade_kwargs."""
    from . import oracle as orc
    nt = n_threads or orc.max_threads()
    # torch's intra-op pool beyond ~32 threads only adds contention on [256 x 10 688] element-wise ops (chunks that stay in the caches: 256 rays is the fastest of 128 .. 4096 on 8 threads) (128 threads: 67 s, 2.7 x slower
    # than 8 threads on the same work): the baseline reports the threads it actually used
    nt = min(nt, 32)
    torch.set_num_threads(nt)
    S = n_samples_x * n_samples_x
    t0 = time.perf_counter()
    # pass A: the sample directions (visibility does not enter them: kernel.cu:507-530 draws the samples before tracing)
    mask = kw['mask']
    NHW = mask.numel()
    a = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n_samples_x, rnd_seed=rnd_seed, n_threads=nt,
                      vis_in=torch.ones(NHW, 2 * S, dtype=torch.uint8), want_dbg=True)
    t1 = time.perf_counter()
    cov = (mask.reshape(-1) > 0).nonzero().reshape(-1)
    dirs = a['dbg'][cov, :, 0:3].reshape(-1, 3).contiguous()
    ro = kw['ro'].reshape(-1, 3)[cov][:, None, :].expand(-1, 2 * S, -1).reshape(-1, 3).contiguous()
    v = shadow_rays_bruteforce_torch(mesh['v_pos'], mesh['t_pos_idx'], ro, dirs)
    vis = torch.ones(NHW, 2 * S, dtype=torch.uint8)
    vis[cov] = v.reshape(-1, 2 * S)
    t2 = time.perf_counter()
    f = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n_samples_x, rnd_seed=rnd_seed, n_threads=nt, vis_in=vis)
    b = None
    t3 = time.perf_counter()
    if diff_grad is not None:
        # the reference re-traces every shadow ray in backward (torch_bindings.cpp:238,266): the torch shadow test runs again
        v2 = shadow_rays_bruteforce_torch(mesh['v_pos'], mesh['t_pos_idx'], ro, dirs)
        vis2 = torch.ones(NHW, 2 * S, dtype=torch.uint8)
        vis2[cov] = v2.reshape(-1, 2 * S)
        b = orc.env_shade(mesh['v_pos'], mesh['t_pos_idx'], **kw, bsdf='pbr', n_samples_x=n_samples_x, rnd_seed=rnd_seed, n_threads=nt,
                          vis_in=vis2, diff_grad=diff_grad, spec_grad=spec_grad)
    t4 = time.perf_counter()
    timing = {'directions_s': t1 - t0, 'torch_shadow_fwd_s': t2 - t1, 'shade_fwd_s': t3 - t2, 'bwd_s': t4 - t3, 'total_s': t4 - t0,
              'covered': int(cov.numel()), 'rays_per_pass': int(cov.numel()) * 2 * S, 'threads': nt}
    if want:
        timing['vis'] = vis
        timing['rays'] = (ro, dirs)
    return f, b, timing
