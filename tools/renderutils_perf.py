"""The reference's two timing harnesses with recorded numbers: pbr_bsdf as torch ops (`use_python=True`) against the HIP op at the three
shapes of render/renderutils/tests/test_perf.py:18-56, forward as there and forward + backward; the bilateral filter as torch ops (the
roll-based formulation of render/optixutils/tests/filter_test.py:30-72, restated in tests/test_gpu_denoiser.py) against the HIP op, forward +
backward at sigma = 2 as filter_test.py:94-112.  usage: python tools/renderutils_perf.py [iterations] [out.md]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import renderutils as ru, optixutils as ou  # noqa: E402

ITR = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
rows = []


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bsdf(batch, res):
    g = torch.Generator().manual_seed(1)
    t = [torch.rand(batch, res, res, 3, generator=g).to(dev).requires_grad_(True) for _ in range(6)]
    go = torch.rand(batch, res, res, 3, generator=g).to(dev)
    for label, kw in (('torch ops', {'use_python': True}), ('HIP', {})):
        fwd = timed(lambda: ru.pbr_bsdf(*t, **kw), ITR)

        def both():
            for x in t:
                x.grad = None
            ru.pbr_bsdf(*t, **kw).backward(go)
        fb = timed(both, max(ITR // 4, 5))
        rows.append(('pbr_bsdf [%d, %d, %d]' % (batch, res, res), label, fwd, fb))


def torch_filter(col, nrm, zdz, sigma):
    """filter_test.py:30-72 (the authors' formulation with torch.roll), weights without gradient as there."""
    eps = 1e-4
    R = 2 * int(__import__('math').ceil(sigma * 2.5)) + 1
    H, W = col.shape[1], col.shape[2]
    ty, tx = torch.meshgrid(torch.arange(H, device=col.device), torch.arange(W, device=col.device), indexing='ij')
    acc, accw = torch.zeros_like(col), torch.zeros_like(col[..., 0:1])
    for y in range(-(R // 2), R // 2 + 1):
        for x in range(-(R // 2), R // 2 + 1):
            with torch.no_grad():
                d2 = float(x * x + y * y)
                w_xy = __import__('math').exp(-d2 / (2.0 * sigma * sigma))
                nt = torch.roll(nrm, (-y, -x), (1, 2))
                wn = torch.pow(torch.clamp((nt * nrm).sum(-1, keepdim=True), min=eps, max=1.0), 128.0)
                zt = torch.roll(zdz, (-y, -x), (1, 2))
                wd = torch.exp(-(torch.abs(zt[..., 0:1] - zdz[..., 0:1]) / torch.clamp(zdz[..., 1:2] * (d2 ** 0.5), min=eps)))
                w = w_xy * wn * wd
                inside = ((tx + x >= 0) & (tx + x < W) & (ty + y >= 0) & (ty + y < H))[None, ..., None]
                w = torch.where(inside, w, torch.zeros_like(w))
            acc = acc + torch.roll(col, (-y, -x), (1, 2)) * w
            accw = accw + w
    return acc / torch.clamp(accw, min=eps)


def bilateral(res, n=1):
    g = torch.Generator().manual_seed(2)
    col = torch.rand(n, res, res, 3, generator=g).to(dev).requires_grad_(True)
    nrm = torch.nn.functional.normalize(torch.rand(n, res, res, 3, generator=g), dim=-1).to(dev)
    zdz = torch.rand(n, res, res, 2, generator=g).to(dev)
    tgt = torch.rand(n, res, res, 3, generator=g).to(dev)

    def run(f):
        col.grad = None
        torch.nn.functional.mse_loss(f(col, nrm, zdz, 2.0), tgt).backward()
    rows.append(('bilateral filter sigma 2 [%d, %d, %d], fwd + bwd' % (n, res, res), 'torch ops (121 rolls)', float('nan'), timed(lambda: run(torch_filter), 3)))
    rows.append(('bilateral filter sigma 2 [%d, %d, %d], fwd + bwd' % (n, res, res), 'HIP', float('nan'), timed(lambda: run(ou.bilateral_denoiser), ITR)))


for b, r in ((1, 512), (16, 512), (1, 2048)):
    bsdf(b, r)
bilateral(1024)
bilateral(512, 8)
lines = ['# The reference\'s timing harnesses on MI355X (`tools/renderutils_perf.py`)', '',
         'render/renderutils/tests/test_perf.py:18-56 (pbr_bsdf, `use_python=True` against the compiled op) and render/optixutils/tests/filter_test.py:94-112',
         '(bilateral filter as torch ops against the compiled op): the reference prints these times and records none.  ms per call, HIP events.', '',
         '| op | path | forward | forward + backward |', '|---|---|---|---|']
for name, label, f, fb in rows:
    lines.append('| %s | %s | %s | %.3f |' % (name, label, '%.3f' % f if f == f else '', fb))
text = '\n'.join(lines) + '\n'
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text)
