"""BVH build alone on the GPU (nothing else running): HIP-event time of nvdr_bvh_build for a mesh, rebuild and refit, on the
caller's stream.  usage: bvh_probe.py [mesh] [subdiv] [reps]   (under rocprofv3 --kernel-trace for the per-kernel table)"""
import os
import sys

os.environ.setdefault('NVDR_TUNING', '1')
os.environ.setdefault('NVDR_ASYNC_BUILD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nvdiffrecmc_amd import optixutils as ou, scene as sc  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else 'bob'
subdiv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device('cuda:0')
m = sc.load_mesh(mesh)
v, t = (sc.subdivide(m['v_pos'], m['t_pos_idx'], subdiv) if subdiv else (m['v_pos'], m['t_pos_idx']))
v, t = v.to(dev), t.to(dev)
ctx = ou.OptiXContext()
for rebuild in (1, 0):
    for _ in range(5):
        ou.optix_build_bvh(ctx, v, t, rebuild=rebuild)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        ou.optix_build_bvh(ctx, v, t, rebuild=rebuild)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print('%s subdiv %d: %d triangles, %s: median %.3f ms, min %.3f, max %.3f (alone on the GPU, caller stream)'
          % (mesh, subdiv, t.shape[0], 'rebuild' if rebuild else 'refit', ms[len(ms) // 2], ms[0], ms[-1]))
ctx.check()
