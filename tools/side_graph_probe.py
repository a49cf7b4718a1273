"""Does a HIP graph replayed on a side stream run beside work on the main stream?  (round 6: the split several-rank schedule replays the
BVH rebuild as a graph of its own on a side stream; the first measurement showed the main stream standing still for it.)
Cases: the rebuild graph on each of several fresh side streams while the main stream runs (a) nothing but two events, (b) a long kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nvdiffrecmc_amd import optixutils as ou, scene as sc  # noqa: E402

dev = torch.device('cuda:0')
m = sc.load_mesh('bob')
v, t = sc.subdivide(m['v_pos'], m['t_pos_idx'], int(os.environ.get('PROBE_SUBDIV', '3')))
v, t = v.to(dev), t.to(dev)
ctx = ou.OptiXContext()
for _ in range(3):
    ou.optix_build_bvh(ctx, v, t, rebuild=1)
ctx.wait_build()
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    ou.optix_build_bvh(ctx, v, t, rebuild=1)
    ctx.wait_build()
ctx.build_joined()
big = torch.zeros(1 << 28, device=dev)
main = torch.cuda.current_stream()


def ms(a, b):
    return a.elapsed_time(b)


def E():
    return torch.cuda.Event(enable_timing=True)


# the graph alone on the main stream
for _ in range(3):
    g1.replay()
torch.cuda.synchronize()
a, b = E(), E()
a.record(); g1.replay(); b.record(); torch.cuda.synchronize()
print('rebuild graph alone on the main stream: %.3f ms' % ms(a, b))
big.add_(1.0); torch.cuda.synchronize()
a, b = E(), E()
a.record(); big.add_(1.0); b.record(); torch.cuda.synchronize()
t_big = ms(a, b)
print('the long kernel alone: %.3f ms' % t_big)
sides = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1) for _ in range(6)]      # (6 normal, 6 high priority)
for k, side in enumerate(sides):
    for mode in ('events only', 'long kernel on main'):
        ev_main, ev_side0, ev_side1 = torch.cuda.Event(), E(), E()
        a, b, c = E(), E(), E()
        torch.cuda.synchronize()
        a.record(main)
        ev_main.record(main)
        side.wait_event(ev_main)
        with torch.cuda.stream(side):
            ev_side0.record(side)
            g1.replay()
            ev_side1.record(side)
        b.record(main)
        if mode != 'events only':
            big.add_(1.0)
        c.record(main)
        torch.cuda.synchronize()
        print('side stream %d, %-20s main a->b %.3f ms, a->c %.3f ms; side graph %.3f ms; a->side end %.3f ms' % (k, mode + ':', ms(a, b), ms(a, c), ms(ev_side0, ev_side1), ms(a, ev_side1)))
# the same work eagerly (no graph) on a side stream
side = sides[0]
for mode in ('events only', 'long kernel on main'):
    ev_main, s0, s1 = torch.cuda.Event(), E(), E()
    a, b, c = E(), E(), E()
    torch.cuda.synchronize()
    a.record(main)
    ev_main.record(main)
    side.wait_event(ev_main)
    with torch.cuda.stream(side):
        s0.record(side)
        ou.optix_build_bvh(ctx, v, t, rebuild=1)
        ctx.wait_build()
        s1.record(side)
    ctx.build_joined()
    b.record(main)
    if mode != 'events only':
        big.add_(1.0)
    c.record(main)
    torch.cuda.synchronize()
    print('EAGER build on side stream 0, %-20s main a->b %.3f ms, a->c %.3f ms; side %.3f ms; a->side end %.3f ms' % (mode + ':', ms(a, b), ms(a, c), ms(s0, s1), ms(a, s1)))
