"""CPU model of the treelet rebuild of csrc/bvh.hip (round 5): Karras LBVH whose maximal subtrees of <= W leaves (inside one block of 256
Morton-sorted leaves) are rebuilt by mutual-nearest-neighbour agglomerative clustering on the surface area of the union, under the eight-wide
SAH-optimal collapse and the walk model of tools/oct_model.py: node steps / box tests / triangle tests per shadow ray.  CPU only:
    python tools/treelet_model.py [mesh] [n_rays]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc
from tools import tree_quality_probe as tq
from tools import oct_model as om
mesh_name = sys.argv[1] if len(sys.argv) > 1 else 'bob'
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
m = sc.load_mesh(mesh_name)
tri = m['v_pos'].numpy().astype(np.float64)[m['t_pos_idx'].numpy()]
ro, rd = tq.shadow_rays(mesh_name, n_rays)
leaf_lo, leaf_hi, g0, gs = om.grid_boxes(tri)
ref = om.brute(tri, ro, rd)
base = None
def report(name, left, right, root, extra=''):
    global base
    lo, hi = om.fit_int(left, right, leaf_lo, leaf_hi, root)
    nodes = om.collapse_dp(left, right, lo, hi, leaf_lo, leaf_hi, gs, root, 1.0, 0.45)
    vis, st, nb, nt, msp = om.walk(nodes, tri, g0, gs, ro, rd)
    if base is None: base = st
    print('  %-40s %6d oct nodes | %.2f node steps (%+5.1f %%), %.1f box tests, %.2f tri tests, %d rays differ %s'
          % (name, len(nodes), st, 100 * (st / base - 1), nb, nt, int((vis != ref).sum()), extra), flush=True)

def build(tri, W, block=None, cap=None, bits=10):
    cen = tri.mean(1)
    vlo, vhi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    keys = tq.morton(cen, vlo, vhi, bits)
    order = np.argsort(keys, kind='stable')
    l0, r0 = tq.karras(keys[order])         # children: >= 0 internal, < 0 -> ~position
    left, right = np.array(l0), np.array(r0)
    n = len(tri)
    bmin, bmax = tri.min(1)[order].astype(np.float32), tri.max(1)[order].astype(np.float32)   # by position
    sys.setrecursionlimit(100000)
    rng = {}
    def ranges(v):
        if v < 0: return (~v, ~v)
        a = ranges(left[v]); b = ranges(right[v]); rng[v] = (min(a[0], b[0]), max(a[1], b[1])); return rng[v]
    ranges(0)
    stats = {'treelets': 0, 'reverted': 0, 'maxdepth': 0, 'rounds': []}
    def ok(v):
        lo, hi = rng[v]
        return hi - lo + 1 <= W and (block is None or lo // block == hi // block)
    def rebuild(v):
        lo, hi = rng[v]
        k = hi - lo + 1
        if k < 3: return
        stats['treelets'] += 1
        cl = [[~p, bmin[p].copy(), bmax[p].copy(), 0] for p in range(lo, hi + 1)]    # ref, lo, hi, depth
        active = list(range(k))
        merges = []
        c = 0
        rounds = 0
        def area(a, b):
            e = np.maximum(a[2], b[2]) - np.minimum(a[1], b[1])
            return np.float32(e[0] * e[1] + e[1] * e[2] + e[2] * e[0])
        while len(active) > 1:
            rounds += 1
            nn = {}
            for i in active:
                best, bj = None, -1
                for j in active:
                    if j == i: continue
                    a = area(cl[i], cl[j])
                    if best is None or a < best: best, bj = a, j
                nn[i] = bj
            pairs = [(i, nn[i]) for i in active if nn[nn[i]] == i and i < nn[i]]
            for rank, (i, p) in enumerate(pairs):
                nid = v if len(active) == 2 else lo + 1 + c + rank
                merges.append((nid, cl[i][0], cl[p][0]))
                cl[i] = [nid, np.minimum(cl[i][1], cl[p][1]), np.maximum(cl[i][2], cl[p][2]), 1 + max(cl[i][3], cl[p][3])]
            c += len(pairs)
            dead = set(p for _, p in pairs)
            active = [a for a in active if a not in dead]
        depth = cl[active[0]][3]
        stats['maxdepth'] = max(stats['maxdepth'], depth); stats['rounds'].append(rounds)
        if cap is not None and depth > cap:
            stats['reverted'] += 1
            return
        ids = sorted(x[0] for x in merges)
        assert ids[-1] <= hi and ids[0] >= lo and len(set(ids)) == k - 1, (ids, lo, hi, v)
        for nid, a, b in merges: left[nid], right[nid] = a, b
    def descend(v):
        if v < 0: return
        if ok(v): rebuild(v); return
        descend(left[v]); descend(right[v])
    descend(0)
    conv = lambda c: c if c >= 0 else ~int(order[~c])
    L = np.array([conv(c) for c in left]); R = np.array([conv(c) for c in right])
    return L, R, 0, stats

print('%s: %d triangles, %d rays' % (mesh_name, len(tri), n_rays))
l, r, root = tq.build_lbvh(tri, 10); report('lbvh', l, r, root)
for W, block, cap in ((64, None, None), (64, 256, None), (64, 256, 20), (32, 256, 20), (128, 256, 24)):
    t0 = time.time(); l, r, root, st = build(tri, W, block, cap)
    report('treelets W=%d block=%s cap=%s' % (W, block, cap), l, r, root, '| %d treelets, %d reverted, max depth %d, rounds avg %.1f max %d (%.0f s)' % (st['treelets'], st['reverted'], st['maxdepth'], np.mean(st['rounds']), max(st['rounds']), time.time() - t0))
