"""Binary-tree builders of tools/tree_quality_probe.py (LBVH, LBVH + rotations, PLOC, binned SAH) under the eight-wide SAH-optimal
collapse and the walk model of tools/oct_model.py: node steps / box tests / triangle tests per shadow ray.  CPU only:
    python tools/tree_builders_under_dp.py [mesh] [n_rays]"""
import os
import sys
import time
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
print('%s: %d triangles, %d shadow rays; eight-wide DP collapse (c_leaf 0.45) + the walk of trace_kernel.h, per ray' % (mesh_name, len(tri), n_rays))
base = None
def report(name, left, right, root, dt):
    global base
    lo, hi = om.fit_int(left, right, leaf_lo, leaf_hi, root)
    nodes = om.collapse_dp(left, right, lo, hi, leaf_lo, leaf_hi, gs, root, 1.0, 0.45)
    vis, st, nb, nt, msp = om.walk(nodes, tri, g0, gs, ro, rd)
    if base is None: base = st
    print('  %-28s %6d oct nodes | %.2f node steps (%+5.1f %%), %.1f box tests, %.2f triangle tests, deepest stack %d, %d rays differ from brute force (build %.1f s)'
          % (name, len(nodes), st, 100 * (st / base - 1), nb, nt, msp, int((vis != ref).sum()), dt), flush=True)
t0 = time.time(); l, r, root = tq.build_lbvh(tri, 10); report('lbvh (as bvh.hip)', l, r, root, time.time() - t0)
for passes in (1, 3):
    t0 = time.time()
    l0, r0, root0 = tq.build_lbvh(tri, 10)
    lo, hi = tq.fit(l0, r0, leaf_lo.astype(np.float64), leaf_hi.astype(np.float64), root0)
    l1, r1, lo, hi = tq.rotate(l0.copy(), r0.copy(), lo, hi, leaf_lo.astype(np.float64), leaf_hi.astype(np.float64), root0, passes)
    report('lbvh + rotations x%d' % passes, l1, r1, root0, time.time() - t0)
for radius in (8, 16, 32):
    t0 = time.time(); l, r, root = tq.build_ploc(tri, radius); report('ploc r=%d' % radius, l, r, root, time.time() - t0)
t0 = time.time(); l, r, root = tq.build_sah(tri, 16); report('binned sah (16 bins)', l, r, root, time.time() - t0)
