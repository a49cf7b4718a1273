"""How much traversal work would a better tree save the PRODUCTION walk?  CPU model of csrc/bvh.hip + csrc/trace_kernel.h:
builds several BVH2s over a mesh (one triangle per leaf), collapses each into the four-slot wide nodes of bvh_widen_kernel
(grandchildren, a leaf child keeps its slot), and runs the kernel's unordered any-hit walk (continue with the first hit slot, push
the other hits, pop the last pushed) over shadow rays that start on the surface (origin = G-buffer position + 1e-3 normal,
direction uniform over the upper hemisphere = the live rays of the ray stream).  Reports box tests and triangle tests per ray.

Trees:  lbvh      30-bit Morton codes of the vertex average + Karras hierarchy  (what bvh.hip builds)
        lbvh63    the same with 21 bits per axis
        rot       lbvh + bottom-up tree rotations that reduce the surface-area cost (Kensler 2008), n passes
        ploc      parallel locally-ordered clustering (Meister & Bittner 2018), search radius r
        sah       binned top-down SAH (16 bins), the quality reference
usage: python tools/tree_quality_probe.py [mesh] [n_rays]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc  # noqa: E402
from oracle import scene_cpu  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------ trees
# a tree = (left, right, lo, hi, root): children >= 0 internal node, < 0: ~triangle; lo/hi [n_internal, 3] boxes of the nodes

def expand_bits(v, bits):
    out = np.zeros_like(v, dtype=np.uint64)
    for b in range(bits):
        out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return out


def morton(cen, lo, hi, bits):
    ext = np.where(hi - lo > 0, hi - lo, 1.0)
    f = np.clip((cen - lo) / ext * (1 << bits), 0, (1 << bits) - 1).astype(np.uint64)
    return (expand_bits(f[:, 0], bits) << np.uint64(2)) | (expand_bits(f[:, 1], bits) << np.uint64(1)) | expand_bits(f[:, 2], bits)


def karras(keys):
    """Hierarchy over SORTED keys (ties broken by position); returns left, right (children: >= 0 internal, < 0 ~leaf position)."""
    n = len(keys)
    kb = 64

    def delta(i, j):
        if j < 0 or j >= n:
            return -1
        a, b = int(keys[i]), int(keys[j])
        if a == b:
            return kb + (32 - (i ^ j).bit_length())
        return kb - (a ^ b).bit_length()
    left = np.zeros(n - 1, dtype=np.int64)
    right = np.zeros(n - 1, dtype=np.int64)
    for i in range(n - 1):
        d = 1 if delta(i, i + 1) > delta(i, i - 1) else -1
        dmin = delta(i, i - d)
        lmax = 2
        while delta(i, i + lmax * d) > dmin:
            lmax *= 2
        length = 0
        t = lmax // 2
        while t >= 1:
            if delta(i, i + (length + t) * d) > dmin:
                length += t
            t //= 2
        j = i + length * d
        dnode = delta(i, j)
        s = 0
        t = (length + 1) // 2
        ll = length
        while True:
            if delta(i, i + (s + t) * d) > dnode:
                s += t
            if t == 1:
                break
            t = (t + 1) // 2
        gamma = i + s * d + min(d, 0)
        left[i] = gamma if min(i, j) != gamma else ~gamma
        right[i] = gamma + 1 if max(i, j) != gamma + 1 else ~(gamma + 1)
    return left, right


def fit(left, right, leaf_lo, leaf_hi, root=0):
    """Boxes of the internal nodes (post-order without recursion)."""
    n = len(left)
    lo = np.zeros((n, 3)); hi = np.zeros((n, 3))
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in (left[v], right[v]):
            if c >= 0:
                stack.append(c)
    for v in reversed(order):
        bl = [leaf_lo[~c] if c < 0 else lo[c] for c in (left[v], right[v])]
        bh = [leaf_hi[~c] if c < 0 else hi[c] for c in (left[v], right[v])]
        lo[v] = np.minimum(bl[0], bl[1]); hi[v] = np.maximum(bh[0], bh[1])
    return lo, hi


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])


def build_lbvh(tri, bits):
    cen = tri.mean(1)
    vlo, vhi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    keys = morton(cen, vlo, vhi, bits)
    order = np.argsort(keys, kind='stable')
    left, right = karras(keys[order])
    # leaves refer to sorted positions: map to triangle ids
    conv = lambda c: c if c >= 0 else ~int(order[~c])
    left = np.array([conv(c) for c in left]); right = np.array([conv(c) for c in right])
    return left, right, 0


def rotate(left, right, lo, hi, leaf_lo, leaf_hi, root, passes):
    """Greedy bottom-up rotations: at every node try to swap one child with a grandchild under the other child when that lowers
    the surface area of the node that changes (the classic four candidates)."""
    n = len(left)
    box = lambda c: (leaf_lo[~c], leaf_hi[~c]) if c < 0 else (lo[c], hi[c])
    for _ in range(passes):
        order, stack = [], [root]
        while stack:
            v = stack.pop(); order.append(v)
            for c in (left[v], right[v]):
                if c >= 0:
                    stack.append(c)
        changed = 0
        for v in reversed(order):
            best = None
            for side in (0, 1):
                a = (left, right)[side][v]          # the child that keeps its place ... no: the child that is SWAPPED down
                b = (right, left)[side][v]          # the other child, whose grandchildren are candidates
                if b < 0:
                    continue
                for gs in (0, 1):
                    g = (left, right)[gs][b]        # grandchild that moves up
                    o = (right, left)[gs][b]        # its sibling stays under b together with a
                    alo, ahi = box(a); olo, ohi = box(o)
                    new_area = area(np.minimum(alo, olo), np.maximum(ahi, ohi))
                    gain = area(lo[b], hi[b]) - new_area
                    if gain > 1e-12 and (best is None or gain > best[0]):
                        best = (gain, side, gs)
            if best is not None:
                _, side, gs = best
                A, B = ((left, right), (right, left))[side]
                a, b = A[v], B[v]
                G = (left, right)[gs]
                g = G[b]
                G[b] = a
                A[v] = g
                o = (right, left)[gs][b]
                alo, ahi = box(a); olo, ohi = box(o)
                lo[b] = np.minimum(alo, olo); hi[b] = np.maximum(ahi, ohi)
                changed += 1
        lo, hi = fit(left, right, leaf_lo, leaf_hi, root)
        if not changed:
            break
    return left, right, lo, hi


def build_ploc(tri, radius, bits=21):
    cen = tri.mean(1)
    vlo, vhi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    order = np.argsort(morton(cen, vlo, vhi, bits), kind='stable')
    n = len(tri)
    ids = [~int(t) for t in order]                              # cluster -> child reference
    clo = tri.min(1)[order].copy(); chi = tri.max(1)[order].copy()
    left, right, nlo, nhi = [], [], [], []
    while len(ids) > 1:
        m = len(ids)
        idx = np.arange(m)
        best = np.full(m, -1); bestc = np.full(m, np.inf)
        for off in range(1, radius + 1):
            for sgn in (-1, 1):
                j = idx + sgn * off
                ok = (j >= 0) & (j < m)
                jj = np.clip(j, 0, m - 1)
                c = area(np.minimum(clo, clo[jj]), np.maximum(chi, chi[jj]))
                c = np.where(ok, c, np.inf)
                upd = c < bestc
                best = np.where(upd, jj, best); bestc = np.where(upd, c, bestc)
        mutual = best[best] == idx
        new_ids, new_lo, new_hi = [], [], []
        for i in range(m):
            j = best[i]
            if mutual[i]:
                if i < j:
                    node = len(left)
                    left.append(ids[i]); right.append(ids[j])
                    l = np.minimum(clo[i], clo[j]); h = np.maximum(chi[i], chi[j])
                    nlo.append(l); nhi.append(h)
                    new_ids.append(node); new_lo.append(l); new_hi.append(h)
            else:
                new_ids.append(ids[i]); new_lo.append(clo[i]); new_hi.append(chi[i])
        ids, clo, chi = new_ids, np.array(new_lo), np.array(new_hi)
    return np.array(left), np.array(right), ids[0]


def build_sah(tri, nbins=16):
    bmin, bmax = tri.min(1), tri.max(1)
    cen = 0.5 * (bmin + bmax)
    left, right = [], []

    def rec(idx):
        if len(idx) == 1:
            return ~int(idx[0])
        cmin, cmax = cen[idx].min(0), cen[idx].max(0)
        ext = cmax - cmin
        best = None
        for ax in range(3):
            if ext[ax] <= 0:
                continue
            b = np.minimum(((cen[idx, ax] - cmin[ax]) / ext[ax] * nbins).astype(int), nbins - 1)
            for s in range(1, nbins):
                L = idx[b < s]; R = idx[b >= s]
                if len(L) == 0 or len(R) == 0:
                    continue
                c = area(bmin[L].min(0), bmax[L].max(0)) * len(L) + area(bmin[R].min(0), bmax[R].max(0)) * len(R)
                if best is None or c < best[0]:
                    best = (c, L, R)
        if best is None:
            h = len(idx) // 2
            L, R = idx[:h], idx[h:]
        else:
            _, L, R = best
        me = len(left)
        left.append(0); right.append(0)
        l, r = rec(L), rec(R)
        left[me], right[me] = l, r
        return me
    sys.setrecursionlimit(100000)
    root = rec(np.arange(len(tri)))
    return np.array(left), np.array(right), root


# ------------------------------------------------------------------------------------------------------------------ the walk

def widen(left, right, lo, hi, leaf_lo, leaf_hi):
    """Four-slot nodes as bvh_widen_kernel derives them: the children of the two children; a leaf child keeps one slot."""
    wide = []
    for v in range(len(left)):
        slots = []
        for c in (left[v], right[v]):
            if c >= 0:
                for g in (left[c], right[c]):
                    slots.append(g)
            else:
                slots.append(c)
        wide.append([(int(g), leaf_lo[~g] if g < 0 else lo[g], leaf_hi[~g] if g < 0 else hi[g]) for g in slots])
    return wide


def walk(wide, root, tri, ro, rd):
    v0 = tri[:, 0]; e1 = tri[:, 1] - v0; e2 = tri[:, 2] - v0
    nb = nt = steps = occ = 0
    for o, d in zip(ro, rd):
        with np.errstate(divide='ignore'):
            inv = 1.0 / d
        cur, stack = root, []
        while True:
            if cur >= 0:
                steps += 1
                nxt = None
                for ref, blo, bhi in wide[cur]:
                    nb += 1
                    t0 = (blo - o) * inv; t1 = (bhi - o) * inv
                    tn = max(np.nanmax(np.minimum(t0, t1)), 0.0); tf = np.nanmin(np.maximum(t0, t1))
                    if tn <= tf:
                        if nxt is None:
                            nxt = ref
                        else:
                            stack.append(ref)
                if nxt is None:
                    if not stack:
                        break
                    nxt = stack.pop()
                cur = nxt
            else:
                k = ~cur
                nt += 1
                p = np.cross(d, e2[k]); det = e1[k] @ p
                hit = False
                if det != 0:
                    tv = o - v0[k]
                    u = (tv @ p) / det
                    q = np.cross(tv, e1[k])
                    vv = (d @ q) / det
                    tt = (e2[k] @ q) / det
                    hit = u >= 0 and vv >= 0 and u + vv <= 1 and tt > 0
                if hit:
                    occ += 1
                    break
                if not stack:
                    break
                cur = stack.pop()
    n = len(ro)
    return nb / n, nt / n, steps / n, occ / n


def sah_cost(left, right, lo, hi, root):
    return float(area(lo, hi).sum() / area(lo[root], hi[root]))


def shadow_rays(mesh_name, n_rays, res=192, seed=3):
    inp = scene_cpu.make_inputs(mesh_name, res, res, 2)
    mask = inp['mask'][0].numpy() > 0
    pos = inp['gb_pos'][0].numpy()[mask].astype(np.float64)
    nrm = inp['gb_normal'][0].numpy()[mask].astype(np.float64)
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(pos), n_rays)
    d = rng.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    n = nrm[pick] / np.maximum(np.linalg.norm(nrm[pick], axis=1, keepdims=True), 1e-20)
    d = np.where((d * n).sum(1, keepdims=True) < 0, -d, d)             # upper hemisphere: the live rays
    return pos[pick] + 1e-3 * n, d


def main():
    mesh_name = sys.argv[1] if len(sys.argv) > 1 else 'bob'
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    m = sc.load_mesh(mesh_name)
    v = m['v_pos'].numpy().astype(np.float64)
    t = m['t_pos_idx'].numpy()
    tri = v[t]
    leaf_lo, leaf_hi = tri.min(1), tri.max(1)
    ro, rd = shadow_rays(mesh_name, n_rays)
    print('%s: %d triangles, %d shadow rays from the surface (GPU counting build on the benchmark stream: 44.5 box / 2.87 triangle '
          'tests per ray)' % (mesh_name, len(t), n_rays))
    trees = []
    t0 = time.time(); l, r, root = build_lbvh(tri, 10); trees.append(('lbvh (30-bit, as bvh.hip)', l, r, root, time.time() - t0))
    t0 = time.time(); l2, r2, root2 = build_lbvh(tri, 21); trees.append(('lbvh63', l2, r2, root2, time.time() - t0))
    for passes in (1, 3):
        t0 = time.time()
        lo, hi = fit(l.copy(), r.copy(), leaf_lo, leaf_hi, root)
        lr, rr, _, _ = rotate(l.copy(), r.copy(), lo, hi, leaf_lo, leaf_hi, root, passes)
        trees.append(('lbvh + rotations x%d' % passes, lr, rr, root, time.time() - t0))
    for radius in (8, 25):
        t0 = time.time(); lp, rp, rootp = build_ploc(tri, radius); trees.append(('ploc r=%d' % radius, lp, rp, rootp, time.time() - t0))
    t0 = time.time(); ls, rs, roots = build_sah(tri); trees.append(('binned sah', ls, rs, roots, time.time() - t0))
    base = None
    for name, left, right, root, bt in trees:
        lo, hi = fit(left, right, leaf_lo, leaf_hi, root)
        wide = widen(left, right, lo, hi, leaf_lo, leaf_hi)
        nb, nt, st, occ = walk(wide, root, tri, ro, rd)
        base = base or nb
        print('  %-28s SAH cost %7.2f | wide walk: %6.2f box tests (%+5.1f %%), %5.2f triangle tests, %5.2f steps per ray, occluded %.3f  (build %.1f s)'
              % (name, sah_cost(left, right, lo, hi, root), nb, 100 * (nb / base - 1), nt, st, occ, bt))


if __name__ == '__main__':
    main()
