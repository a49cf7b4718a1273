"""How much traversal work would a SAH tree save over the LBVH?  Builds a binned-SAH BVH2 (1 triangle per leaf) for a mesh
on the CPU and simulates the any-hit traversal (near child first) on the same random rays tools/trace_probe.py uses;
compare node visits / triangle tests per ray with the LBVH numbers measured on the GPU."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc

def build_sah(v, t, nbins=16, lbvh=False):
    tri = v[t]                                   # [T,3,3]
    bmin, bmax = tri.min(1), tri.max(1)
    cen = 0.5 * (bmin + bmax)
    nodes = []                                   # (lmin,lmax,rmin,rmax,left,right); child<0: ~tri
    def area(lo, hi):
        d = np.maximum(hi - lo, 0); return 2 * (d[0]*d[1] + d[1]*d[2] + d[2]*d[0])
    def rec(idx):
        if len(idx) == 1:
            return ~int(idx[0])
        cmin, cmax = cen[idx].min(0), cen[idx].max(0)
        ext = cmax - cmin
        best = None
        for ax in range(3):
            if ext[ax] <= 0: continue
            b = np.minimum(((cen[idx, ax] - cmin[ax]) / ext[ax] * nbins).astype(int), nbins - 1)
            for s in range(1, nbins):
                L = idx[b < s]; R = idx[b >= s]
                if len(L) == 0 or len(R) == 0: continue
                c = area(bmin[L].min(0), bmax[L].max(0)) * len(L) + area(bmin[R].min(0), bmax[R].max(0)) * len(R)
                if best is None or c < best[0]: best = (c, L, R)
        if best is None:
            h = len(idx) // 2; L, R = idx[:h], idx[h:]
        else:
            _, L, R = best
        me = len(nodes); nodes.append(None)
        l, r = rec(L), rec(R)
        nodes[me] = (bmin[L].min(0), bmax[L].max(0), bmin[R].min(0), bmax[R].max(0), l, r)
        return me
    sys.setrecursionlimit(100000)
    rec(np.arange(len(t)))
    return nodes, tri

def traverse(nodes, tri, ro, rd):
    nv = nt = 0
    occl = 0
    v0 = tri[:, 0]; e1 = tri[:, 1] - v0; e2 = tri[:, 2] - v0
    for o, d in zip(ro, rd):
        inv = 1.0 / d
        stack = []; cur = 0
        while True:
            if cur >= 0:
                n = nodes[cur]; nv += 1
                hit = []
                for lo, hi, c in ((n[0], n[1], n[4]), (n[2], n[3], n[5])):
                    t0 = (lo - o) * inv; t1 = (hi - o) * inv
                    tn = max(np.minimum(t0, t1).max(), 0.0); tf = np.maximum(t0, t1).min()
                    if tn <= tf: hit.append((tn, c))
                if len(hit) == 2:
                    hit.sort(key=lambda x: x[0]); stack.append(hit[1][1]); cur = hit[0][1]
                elif len(hit) == 1: cur = hit[0][1]
                elif stack: cur = stack.pop()
                else: break
            else:
                k = ~cur; nt += 1
                p = np.cross(d, e2[k]); det = e1[k] @ p
                if det != 0:
                    tv = o - v0[k]; u = (tv @ p) / det; q = np.cross(tv, e1[k]); vv = (d @ q) / det; tt = (e2[k] @ q) / det
                    if u >= 0 and vv >= 0 and u + vv <= 1 and tt > 0:
                        occl += 1; break
                if stack: cur = stack.pop()
                else: break
    return nv, nt, occl

m = sc.load_mesh(os.environ.get('PROBE_MESH', 'bob'))
v = m['v_pos'].numpy().astype(np.float64); t = m['t_pos_idx'].numpy()
t0 = time.time(); nodes, tri = build_sah(v, t); print('sah build', time.time() - t0, 'nodes', len(nodes))
R = 1 << 22
g = torch.Generator().manual_seed(1)
ro = (torch.randn(R, 3, generator=g) * 0.25).numpy().astype(np.float64)
rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).numpy().astype(np.float64)
K = 4000
nv, nt, oc = traverse(nodes, tri, ro[:K], rd[:K])
print('SAH: node visits/ray %.2f (box tests %.2f)  tri tests/ray %.2f  occluded %.3f' % (nv / K, 2 * nv / K, nt / K, oc / K))
print('LBVH on GPU (same rays): box tests/ray 36.4, tri tests/ray 1.41, occluded 0.563')
