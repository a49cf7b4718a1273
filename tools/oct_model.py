"""CPU model of the round-3 shadow-ray walk: the eight-wide collapse (round 3: bvh_oct_build_kernel, greedy; `dp`: the SAH-optimal one of csrc/bvh.hip) over the LBVH of
tools/tree_quality_probe.py, the 8-bit node-local quantisation, and the unordered group walk of csrc/trace_kernel.h with deferred
triangle tests -- in numpy, float32 where the kernel computes in float32.  Prints what the design costs per ray (node steps, box
tests, triangle tests, stack depth) beside the four-slot walk of round 2, and checks its visibility against brute force.
usage: python tools/oct_model.py [mesh] [n_rays]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc  # noqa: E402
from tools import tree_quality_probe as tq  # noqa: E402

GRID_MAX = 65531.0


def grid_boxes(tri):
    """16-bit leaf boxes as bvh_grid_kernel / bvh_fit_kernel make them."""
    v = tri.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    scale = max((hi - lo).max(), np.abs(lo).max(), np.abs(hi).max())
    pad = 1e-5 * scale
    g0, g1 = lo - 2 * pad, hi + 2 * pad
    gs = GRID_MAX / np.maximum(g1 - g0, 1e-6 * scale + 1e-30)
    tmin, tmax = tri.min(1) - pad, tri.max(1) + pad
    qlo = np.clip(np.floor((tmin - g0) * gs + 2.0).astype(np.int64) - 1, 0, 65535)
    qhi = np.clip(np.ceil((tmax - g0) * gs + 2.0).astype(np.int64) + 1, 0, 65535)
    return qlo, qhi, g0, gs


def fit_int(left, right, leaf_lo, leaf_hi, root=0):
    n = len(left)
    lo = np.zeros((n, 3), dtype=np.int64)
    hi = np.zeros((n, 3), dtype=np.int64)
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
        lo[v] = np.minimum(bl[0], bl[1])
        hi[v] = np.maximum(bh[0], bh[1])
    return lo, hi


def collapse(left, right, lo, hi, leaf_lo, leaf_hi, gs, root=0):
    """-> list of oct nodes: dict(org, e, n_int, n_leaf, cbase, tris (list of triangle ids), qlo [n,3], qhi [n,3])"""
    w = 1.0 / gs
    box = lambda c: (leaf_lo[~c], leaf_hi[~c]) if c < 0 else (lo[c], hi[c])
    nodes, task = [None], [root]
    m = 0
    while m < len(task):
        b = task[m]
        slots = [left[b], right[b]]
        while len(slots) < 8:
            best, best_area = -1, -1.0
            for k, c in enumerate(slots):
                if c < 0:
                    continue
                l, h = box(c)
                e = (h - l) * w
                a = e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
                if a > best_area:
                    best, best_area = k, a
            if best < 0:
                break
            c = slots[best]
            slots[best] = left[c]
            slots.append(right[c])
        ints = [c for c in slots if c >= 0]
        leaves = [c for c in slots if c < 0]
        order = ints + leaves
        bl = np.array([box(c)[0] for c in order])
        bh = np.array([box(c)[1] for c in order])
        org = bl.min(0)
        ext = bh.max(0) - org
        e = np.zeros(3, dtype=np.int64)
        for ax in range(3):
            while ((ext[ax] + (1 << e[ax]) - 1) >> e[ax]) > 255:
                e[ax] += 1
        qlo = (bl - org) >> e
        qhi = ((bh - org) + (1 << e) - 1) >> e
        assert qhi.max() <= 255 and qlo.min() >= 0
        cbase = len(task)
        task.extend(ints)
        nodes.extend([None] * len(ints))
        nodes[m] = dict(org=org, e=e, n_int=len(ints), n_leaf=len(leaves), cbase=cbase, tris=[~c for c in leaves], qlo=qlo, qhi=qhi)
        m += 1
    return nodes


def walk(nodes, tri, g0, gs, ro, rd):
    f32 = np.float32
    v0 = tri[:, 0]
    e1 = tri[:, 1] - v0
    e2 = tri[:, 2] - v0
    n_step = n_box = n_tri = 0
    max_sp = 0
    vis = np.ones(len(ro), dtype=np.uint8)
    for r, (o, d) in enumerate(zip(ro.astype(f32), rd.astype(f32))):
        with np.errstate(divide='ignore', over='ignore'):
            inv = np.clip(f32(1.0) / (d * gs.astype(f32)), f32(-1e30), f32(1e30)).astype(f32)
        noi = (-(((o - g0.astype(f32)) * gs.astype(f32)) + f32(2.0)) * inv).astype(f32)
        gbase, gbits, stack = 0, 1, []
        leaves = []
        while gbits or stack:
            if not gbits:
                gbase, gbits = stack.pop()
            k = (gbits & -gbits).bit_length() - 1
            gbits &= gbits - 1
            nd = nodes[gbase + k]
            n_step += 1
            a = (inv * np.exp2(nd['e']).astype(f32)).astype(f32)
            b = (nd['org'].astype(f32) * inv + noi).astype(f32)
            tl = (nd['qlo'].astype(f32) * a + b).astype(f32)
            th = (nd['qhi'].astype(f32) * a + b).astype(f32)
            tn = np.where(inv < 0, th, tl)
            tf = np.where(inv < 0, tl, th)
            tnear = np.maximum(np.maximum(tn[:, 0], tn[:, 1]), np.maximum(tn[:, 2], f32(0)))
            tfar = np.minimum(np.minimum(tf[:, 0], tf[:, 1]), tf[:, 2])
            hit = ~np.signbit(tfar - tnear)
            n_box += len(hit)
            hi_bits = sum(1 << j for j in range(nd['n_int']) if hit[j])
            for j in range(nd['n_leaf']):
                if hit[nd['n_int'] + j]:
                    leaves.append(nd['tris'][j])
            if hi_bits:
                if gbits:
                    stack.append((gbase, gbits))
                    max_sp = max(max_sp, len(stack))
                gbase, gbits = nd['cbase'], hi_bits
        # deferred triangle tests (all of them: the model does not end a walk early, so n_step is the unoccluded-ray cost
        # plus the speculative steps of occluded rays -- an upper bound of what the kernel does)
        od, dd = o.astype(np.float64), d.astype(np.float64)
        for k in leaves:
            n_tri += 1
            p = np.cross(dd, e2[k])
            det = e1[k] @ p
            if det == 0:
                continue
            tv = od - v0[k]
            u = (tv @ p) / det
            q = np.cross(tv, e1[k])
            vv = (dd @ q) / det
            tt = (e2[k] @ q) / det
            if u >= 0 and vv >= 0 and u + vv <= 1 and tt > 0:
                vis[r] = 0
    n = len(ro)
    return vis, n_step / n, n_box / n, n_tri / n, max_sp


def brute(tri, ro, rd):
    v0 = tri[:, 0]
    e1 = tri[:, 1] - v0
    e2 = tri[:, 2] - v0
    vis = np.ones(len(ro), dtype=np.uint8)
    for r, (o, d) in enumerate(zip(ro, rd)):
        p = np.cross(d[None], e2)
        det = (e1 * p).sum(1)
        ok = det != 0
        det = np.where(ok, det, 1.0)
        tv = o[None] - v0
        u = (tv * p).sum(1) / det
        q = np.cross(tv, e1)
        v = (q @ d) / det
        t = (e2 * q).sum(1) / det
        if (ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)).any():
            vis[r] = 0
    return vis


def main():
    mesh_name = sys.argv[1] if len(sys.argv) > 1 else 'bob'
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    m = sc.load_mesh(mesh_name)
    v = m['v_pos'].numpy().astype(np.float64)
    t = m['t_pos_idx'].numpy()
    tri = v[t]
    ro, rd = tq.shadow_rays(mesh_name, n_rays)
    left, right, root = tq.build_lbvh(tri, 10)
    leaf_lo, leaf_hi, g0, gs = grid_boxes(tri)
    lo, hi = fit_int(left, right, leaf_lo, leaf_hi, root)
    nodes = collapse(left, right, lo, hi, leaf_lo, leaf_hi, gs, root)
    fill = np.mean([nd['n_int'] + nd['n_leaf'] for nd in nodes])
    print('%s: %d triangles -> %d oct nodes (%.2f slots used per node, %d KB instead of %d KB of four-slot nodes)'
          % (mesh_name, len(t), len(nodes), fill, len(nodes) * 64 // 1024, (len(t) - 1) * 64 // 1024))
    vis, st, nb, nt, msp = walk(nodes, tri, g0, gs, ro, rd)
    ref = brute(tri, ro, rd)
    print('oct walk: %.2f node steps, %.1f box tests, %.2f deferred triangle tests per ray, deepest stack %d; occluded %.3f; '
          '%d of %d rays differ from brute force' % (st, nb, nt, msp, 1 - vis.mean(), int((vis != ref).sum()), len(ref)))
    flo, fhi = tq.fit(left, right, tri.min(1), tri.max(1), root)
    wide = tq.widen(left, right, flo, fhi, tri.min(1), tri.max(1))
    nb4, nt4, st4, occ4 = tq.walk(wide, root, tri, ro, rd)
    print('four-slot walk of round 2 (exact boxes): %.2f steps, %.1f box tests, %.2f triangle tests per ray' % (st4, nb4, nt4))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "dp"):
    main()


# ---------------------------------------------------------------------------------------------------------------------
# SAH-optimal collapse (Ylitie, Karras, Laine 2017, section 4.1) as a dynamic programme over the binary tree:
#   c(n, i) = cheapest way to represent the subtree of n with AT MOST i slots of its ancestor's wide node, i = 1..7
#   c(n, 1) = min(leaf cost if n is a leaf, A_n * C_NODE + c_root(n));  c_root(n) = min_k c(l, k) + c(r, 8 - k)
#   c(n, i) = min(c(n, 1), min_k c(l, k) + c(r, i - k))                  (i > 1: distribute the slots over the two children)
C_NODE, C_LEAF = 1.0, 0.45


def collapse_dp(left, right, lo, hi, leaf_lo, leaf_hi, gs, root=0, c_node=C_NODE, c_leaf=C_LEAF):
    w = 1.0 / gs
    n = len(left)

    def area_of(l, h):
        e = (h - l) * w
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in (left[v], right[v]):
            if c >= 0:
                stack.append(c)
    INF = 1e30
    cost = np.full((n, 8), INF)          # cost[v][i], i = 1..7
    split = np.zeros((n, 8), dtype=np.int64)   # slots given to the left child for budget i (0 = keep v as ONE internal slot)
    rsplit = np.zeros(n, dtype=np.int64)       # slots given to the left child when v is the root of a wide node
    leaf_cost = lambda c: area_of(leaf_lo[~c], leaf_hi[~c]) * c_leaf

    def cget(c, i):
        return leaf_cost(c) if c < 0 else cost[c][i]
    for v in reversed(order):
        l, r = left[v], right[v]
        best, bk = INF, 1
        for k in range(1, 8):
            cc = cget(l, k) + cget(r, 8 - k)
            if cc < best:
                best, bk = cc, k
        rsplit[v] = bk
        a = area_of(lo[v], hi[v])
        cost[v][1] = a * c_node + best
        split[v][1] = 0
        for i in range(2, 8):
            best_i, bk_i = cost[v][1], 0
            for k in range(1, i):
                cc = cget(l, k) + cget(r, i - k)
                if cc < best_i:
                    best_i, bk_i = cc, k
            cost[v][i] = best_i
            split[v][i] = bk_i
    box = lambda c: (leaf_lo[~c], leaf_hi[~c]) if c < 0 else (lo[c], hi[c])
    nodes, task = [None], [root]
    m = 0
    while m < len(task):
        b = task[m]
        slots = []

        def emit(c, budget):
            if c < 0 or split[c][budget] == 0:
                slots.append(c)
            else:
                k = split[c][budget]
                emit(left[c], k)
                emit(right[c], budget - k)
        emit(left[b], rsplit[b])
        emit(right[b], 8 - rsplit[b])
        assert len(slots) <= 8
        ints = [c for c in slots if c >= 0]
        leaves = [c for c in slots if c < 0]
        order2 = ints + leaves
        bl = np.array([box(c)[0] for c in order2])
        bh = np.array([box(c)[1] for c in order2])
        org = bl.min(0)
        ext = bh.max(0) - org
        e = np.zeros(3, dtype=np.int64)
        for ax in range(3):
            while ((ext[ax] + (1 << e[ax]) - 1) >> e[ax]) > 255:
                e[ax] += 1
        qlo = (bl - org) >> e
        qhi = ((bh - org) + (1 << e) - 1) >> e
        cbase = len(task)
        task.extend(ints)
        nodes.extend([None] * len(ints))
        nodes[m] = dict(org=org, e=e, n_int=len(ints), n_leaf=len(leaves), cbase=cbase, tris=[~c for c in leaves], qlo=qlo, qhi=qhi)
        m += 1
    return nodes


def main_dp():
    mesh_name = sys.argv[2] if len(sys.argv) > 2 else 'bob'
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    m = sc.load_mesh(mesh_name)
    v = m['v_pos'].numpy().astype(np.float64)
    t = m['t_pos_idx'].numpy()
    tri = v[t]
    ro, rd = tq.shadow_rays(mesh_name, n_rays)
    left, right, root = tq.build_lbvh(tri, 10)
    leaf_lo, leaf_hi, g0, gs = grid_boxes(tri)
    lo, hi = fit_int(left, right, leaf_lo, leaf_hi, root)
    ref = brute(tri, ro, rd)
    for c_leaf in (0.3, 0.45, 0.7, 1.0):
        nodes = collapse_dp(left, right, lo, hi, leaf_lo, leaf_hi, gs, root, 1.0, c_leaf)
        fill = np.mean([nd['n_int'] + nd['n_leaf'] for nd in nodes])
        vis, st, nb, nt, msp = walk(nodes, tri, g0, gs, ro, rd)
        print('DP collapse c_leaf=%.2f: %d oct nodes (%.2f slots per node): %.2f node steps, %.1f box tests, %.2f triangle tests per ray, '
              'deepest stack %d, %d rays differ' % (c_leaf, len(nodes), fill, st, nb, nt, msp, int((vis != ref).sum())))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'dp':
    main_dp()
