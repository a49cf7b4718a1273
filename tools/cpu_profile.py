"""Where does the HOST time of one training iteration go?  cProfile over 30 steps of the benchmark iteration
(GPU work is asynchronous: with the device ahead of the host, wall time per step = host time per step)."""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd.trainer import DirectLightingStep
st = DirectLightingStep('bob', 512, 8, view=0, n_views=8, device='cuda:0', retrace_backward=True)
for _ in range(5): st.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): st.step()
t1 = time.perf_counter()          # host-side issue time only
torch.cuda.synchronize()
t2 = time.perf_counter()
print('30 steps: host issue %.2f ms/step, until GPU done %.2f ms/step' % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30): st.step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr, stream=sys.stdout).sort_stats('cumulative')
ps.print_stats(35)
