"""Box tests per ray of EXACTLY axis-parallel rays vs random rays (binary walk of the library in csrc/build, counting build) and the
time of the production kernel on the axis-parallel rays, for the current library and for every variant under csrc/build/variants --
e.g. the uncapped inverse direction of rounds 1-2:  tools/build_variants.sh "nocap:-DNVDR_INV_CAP=0"  (a variant rebuilds
env_shade.hip only, i.e. the production kernel; the binary walk of bvh.hip stays the default build).
profiles/r02_slow_mode.md, section "Root cause"."""
import glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _lib, _build, scene as sc
from nvdiffrecmc_amd import optixutils as ou

subdiv = int(os.environ.get('PROBE_SUBDIV', '1'))
mesh = sc.load_mesh('bob')
v, t = (mesh['v_pos'], mesh['t_pos_idx']) if subdiv == 0 else sc.subdivide(mesh['v_pos'], mesh['t_pos_idx'], subdiv)
dev = 'cuda:0'
g = torch.Generator().manual_seed(11)
n = 6000
ro = (v[torch.randint(0, v.shape[0], (n,), generator=g)] * 1.001).contiguous().to(dev)       # just outside the surface
dirs = torch.tensor([[0.0, 1.0, -0.0], [-0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [-1.0, -0.0, 0.0], [0.0, -0.0, 1.0], [-0.0, 0.0, -1.0]])
rd = dirs.repeat(n // 6, 1).contiguous().to(dev)
rr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
base = _build.LIB
for tag, path in [('current', base)] + [(p.split('.so.')[-1], p) for p in sorted(glob.glob(os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.*')))]:
    _lib._lib = None
    _build.LIB = path
    _lib.load()
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, v.to(dev), t.to(dev), rebuild=1)
    _, c_axis = ou.trace_visibility(ctx, ro, rd, count=True)
    _, c_rand = ou.trace_visibility(ctx, ro, rr, count=True)
    ou.trace_visibility_wide(ctx, ro, rd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ou.trace_visibility_wide(ctx, ro, rd); e1.record(); torch.cuda.synchronize()
    print('%-8s %d triangles: %8.0f box tests per axis-parallel ray, %6.0f per random ray; production kernel on the %d axis-parallel rays: %.3f ms'
          % (tag, t.shape[0], c_axis[0].item() / n, c_rand[0].item() / n, n, e0.elapsed_time(e1)))
_build.LIB = base
