"""CPU model: does the ORDER in which an any-hit walk visits the internal children of an eight-wide node matter?
The kernel (csrc/trace_kernel.h) takes them in slot order (= Morton order of the collapse).  Here the same tree (tools/oct_model.py,
SAH-optimal collapse) is walked with the triangle tests run when their leaf box is hit (the kernel defers them by a few steps), and
the internal children taken in slot order / reverse / nearest entry first / farthest first / slot order flipped by the sign of the
ray's dominant direction axis.  Prints node steps per ray, for occluded and unoccluded rays.
usage: python tools/order_model.py [mesh] [n_rays]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc  # noqa: E402
from tools import tree_quality_probe as tq  # noqa: E402
from tools import oct_model as om  # noqa: E402


def walk(nodes, tri, g0, gs, ro, rd, order):
    f32 = np.float32
    v0 = tri[:, 0]
    e1 = tri[:, 1] - v0
    e2 = tri[:, 2] - v0
    steps = np.zeros(len(ro))
    vis = np.ones(len(ro), dtype=np.uint8)
    for r, (o, d) in enumerate(zip(ro.astype(f32), rd.astype(f32))):
        with np.errstate(divide='ignore', over='ignore'):
            inv = np.clip(f32(1.0) / (d * gs.astype(f32)), f32(-1e30), f32(1e30)).astype(f32)
        noi = (-(((o - g0.astype(f32)) * gs.astype(f32)) + f32(2.0)) * inv).astype(f32)
        od, dd = o.astype(np.float64), d.astype(np.float64)
        flip = d[np.argmax(np.abs(d))] < 0
        stack = [[0]]                       # stack of lists of node indices still to visit (a group), visited from the END of the list
        done = False
        while stack and not done:
            grp = stack[-1]
            if not grp:
                stack.pop()
                continue
            nd = nodes[grp.pop()]
            steps[r] += 1
            a = (inv * np.exp2(nd['e']).astype(f32)).astype(f32)
            b = (nd['org'].astype(f32) * inv + noi).astype(f32)
            tl = (nd['qlo'].astype(f32) * a + b).astype(f32)
            th = (nd['qhi'].astype(f32) * a + b).astype(f32)
            tn = np.where(inv < 0, th, tl)
            tf = np.where(inv < 0, tl, th)
            tnear = np.maximum(np.maximum(tn[:, 0], tn[:, 1]), np.maximum(tn[:, 2], f32(0)))
            tfar = np.minimum(np.minimum(tf[:, 0], tf[:, 1]), tf[:, 2])
            hit = ~np.signbit(tfar - tnear)
            for j in range(nd['n_leaf']):
                if hit[nd['n_int'] + j]:
                    k = nd['tris'][j]
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
                        done = True
                        break
            if done:
                break
            kids = [j for j in range(nd['n_int']) if hit[j]]
            if kids:
                if order == 'asc':
                    seq = kids
                elif order == 'desc':
                    seq = kids[::-1]
                elif order == 'near':
                    seq = sorted(kids, key=lambda j: tnear[j])
                elif order == 'far':
                    seq = sorted(kids, key=lambda j: -tnear[j])
                elif order == 'sign':
                    seq = kids[::-1] if flip else kids
                elif order == 'small':       # smallest box first (short interval inside the box)
                    seq = sorted(kids, key=lambda j: tfar[j] - tnear[j])
                elif order == 'large':
                    seq = sorted(kids, key=lambda j: -(tfar[j] - tnear[j]))
                stack.append([nd['cbase'] + j for j in seq[::-1]])      # visited from the end
    return vis, steps


def main():
    mesh_name = sys.argv[1] if len(sys.argv) > 1 else 'bob'
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    m = sc.load_mesh(mesh_name)
    v = m['v_pos'].numpy().astype(np.float64)
    t = m['t_pos_idx'].numpy()
    tri = v[t]
    ro, rd = tq.shadow_rays(mesh_name, n_rays)
    left, right, root = tq.build_lbvh(tri, 10)
    leaf_lo, leaf_hi, g0, gs = om.grid_boxes(tri)
    lo, hi = om.fit_int(left, right, leaf_lo, leaf_hi, root)
    nodes = om.collapse_dp(left, right, lo, hi, leaf_lo, leaf_hi, gs, root, 1.0, 0.45)
    ref = om.brute(tri, ro, rd)
    print('%s, %d rays, %.1f %% occluded' % (mesh_name, n_rays, 100.0 * (1 - ref.mean())))
    for order in ('asc', 'desc', 'sign', 'near', 'far', 'small', 'large'):
        vis, steps = walk(nodes, tri, g0, gs, ro, rd, order)
        assert (vis == ref).all()
        print('  %-6s node steps per ray %.2f   (occluded rays %.2f, unoccluded %.2f)' % (order, steps.mean(), steps[ref == 0].mean(), steps[ref == 1].mean()))


if __name__ == '__main__':
    main()
