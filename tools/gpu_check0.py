"""First GPU contact: toolchain/runtime link check, detmath bit-exactness, BVH build + trace vs numpy brute force."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _build
lib = ctypes.CDLL(_build.LIB)
lib.nvdr_last_error.restype = ctypes.c_char_p
def chk(rc, what):
    assert rc == 0, (what, rc, lib.nvdr_last_error())
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = torch.device('cuda:0')
print('device', torch.cuda.get_device_name(0))

# ---- detmath: device vs host (host = the same header compiled with gcc)
import subprocess, tempfile
src = r'''
#include "nvdr_detmath.h"
void run(int op, const float* x, const float* y, long n, float* o){ for(long i=0;i<n;i++){ float s,c; if(op<2){ nvdr_sincosf(x[i],&s,&c); o[i]=op?c:s;} else if(op==2) o[i]=nvdr_acosf(x[i]); else o[i]=nvdr_atan2f(x[i],y[i]); } }
'''
td = tempfile.mkdtemp()
open(td + '/h.c', 'w').write(src)
subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', '-I', os.path.join(os.path.dirname(_build.CSRC), '..', 'include'), td + '/h.c', '-o', td + '/h.so', '-lm'])
host = ctypes.CDLL(td + '/h.so')
n = 1 << 22
rng = np.random.default_rng(0)
for op, name in enumerate(['sin', 'cos', 'acos', 'atan2']):
    if op < 2:
        x = ((rng.random(n) * 2 - 1) * 4 * np.pi).astype(np.float32); y = x
    elif op == 2:
        x = (rng.random(n) * 2 - 1).astype(np.float32); x[:4] = [-1, 1, 0.5, -0.5]; y = x
    else:
        x = (rng.random(n) * 2 - 1).astype(np.float32); y = (rng.random(n) * 2 - 1).astype(np.float32)
    o_h = np.empty(n, np.float32)
    host.run(ctypes.c_int(op), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(n), o_h.ctypes.data_as(ctypes.c_void_p))
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    od = torch.empty(n, device=dev)
    chk(lib.nvdr_test_detmath(ctypes.c_int(op), P(xd), P(yd), ctypes.c_int64(n), P(od), S()), 'detmath')
    o_d = od.cpu().numpy()
    nd = int((o_d.view(np.uint32) != o_h.view(np.uint32)).sum())
    print('detmath', name, 'bit mismatches device vs host:', nd, 'of', n)

# ---- BVH on bob
m = np.load(os.path.join(os.path.dirname(_build.ROOT + '/x'), 'assets', 'bob.npz'))
v = torch.from_numpy(m['v_pos']).to(dev).contiguous(); t = torch.from_numpy(m['t_pos_idx']).to(dev).contiguous()
ctx = ctypes.c_void_p()
chk(lib.nvdr_ctx_create(ctypes.byref(ctx), 0), 'ctx')
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    chk(lib.nvdr_bvh_build(ctx, P(v), ctypes.c_int64(v.shape[0]), P(t), ctypes.c_int64(t.shape[0]), 1, S()), 'build')
    torch.cuda.synchronize(); print('bvh build ms', (time.perf_counter() - t0) * 1e3)
class Info(ctypes.Structure):
    _fields_ = [('n_tris', ctypes.c_int64), ('n_nodes', ctypes.c_int64), ('height', ctypes.c_int32), ('root', ctypes.c_int32), ('mn', ctypes.c_float * 3), ('mx', ctypes.c_float * 3)]
info = Info()
chk(lib.nvdr_bvh_info_get(ctx, ctypes.byref(info), S()), 'info')
print('bvh: tris', info.n_tris, 'nodes', info.n_nodes, 'height', info.height, 'aabb', list(info.mn), list(info.mx))

# random rays from points on a sphere of radius 0.6 around the mesh center + surface-ish origins
R = 1 << 22
g = torch.Generator(device='cpu').manual_seed(1)
ro = (torch.randn(R, 3, generator=g) * 0.25).to(dev)
rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
vis = torch.empty(R, dtype=torch.uint8, device=dev)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
chk(lib.nvdr_trace_visibility(ctx, P(ro), P(rd), ctypes.c_int64(R), P(vis), P(cnt), S()), 'trace')
torch.cuda.synchronize()
print('visible fraction', vis.float().mean().item(), 'box tests/ray', cnt[0].item() / R, 'tri tests/ray', cnt[1].item() / R)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    chk(lib.nvdr_trace_visibility(ctx, P(ro), P(rd), ctypes.c_int64(R), P(vis), None, S()), 'trace')
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('trace %.3f ms  %.2f Grays/s' % (dt * 1e3, R / dt / 1e9))

# brute force check on a subset (numpy, float32, same predicate order is NOT replicated here: loose check)
K = 2000
o = ro[:K].cpu().numpy().astype(np.float64); d = rd[:K].cpu().numpy().astype(np.float64)
V = m['v_pos'].astype(np.float64); T = m['t_pos_idx']
v0 = V[T[:, 0]]; e1 = V[T[:, 1]] - v0; e2 = V[T[:, 2]] - v0
occ = np.zeros(K, bool)
for i in range(K):
    p = np.cross(d[i], e2); det = (e1 * p).sum(-1)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = 1.0 / det; tv = o[i] - v0; u = (tv * p).sum(-1) * inv; q = np.cross(tv, e1); vv = (q * d[i]).sum(-1) * inv; tt = (q * e2).sum(-1) * inv
    occ[i] = np.any((u >= 0) & (vv >= 0) & (u + vv <= 1) & (tt > 0) & (tt < 1e16))
gpu_occ = vis[:K].cpu().numpy() == 0
print('brute-force (fp64) mismatches on', K, 'rays:', int((occ != gpu_occ).sum()))
chk(lib.nvdr_ctx_destroy(ctx), 'destroy')
print('OK')
