#!/usr/bin/env python3
"""Model of the shading kernels' light-sample queue (env_shade_queue_kernel): how many of a pixel's 64 light samples are live (above the
shading normal's horizon) under the importance distribution of the benchmark's probe, and how many 64-lane passes per pixel the queue
needs for ring sizes 2..8 against pairing two pixels, next-fit packing without splitting a pixel, and the ideal.  CPU only:
    python tools/live_queue_model.py"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import scene as sc
m = sc.load_mesh('bob')
base = sc.env_map('E1', 256)
pdf, rows, cols = sc.light_tables(base)
H, W = base.shape[:2]
# importance-sample texels ~ pdf (same distribution as the kernel's CDF inversion), 64 stratified-ish samples per pixel
p = pdf.flatten().double().numpy(); p = p / p.sum()
rng = np.random.default_rng(0)
# pixel normals: vertex normals of triangles facing the camera of view 0 (proxy)
mv, _, campos = sc.camera(0, 8)
vn = m['v_nrm'].numpy(); vp = m['v_pos'].numpy()
cam = campos.numpy()
facing = ((cam[None, :] - vp) * vn).sum(1) > 0
N = vn[facing]
N = N[rng.integers(0, len(N), 20000)]
cnt = []
idx_all = rng.choice(H * W, size=(len(N), 64), p=p)
ty, tx = idx_all // W, idx_all % W
# direction of a texel (lat-long, y up): theta = (ty+.5)/H*pi, phi = (tx+.5)/W*2pi - pi  (convention differences only rotate the map)
th = (ty + 0.5) / H * math.pi; ph = (tx + 0.5) / W * 2 * math.pi - math.pi
d = np.stack([np.sin(th) * np.sin(ph), np.cos(th), -np.sin(th) * np.cos(ph)], -1)
live = (d * N[:, None, :]).sum(-1) > 0
n = live.sum(1)
print('mean live A per pixel', n.mean(), 'hist (bins of 8):', np.histogram(n, bins=range(0, 73, 8))[0] / len(n))
pairs = n[0::2] + n[1::2]
print('random pairs fitting one pass:', (pairs <= 64).mean())
# adjacent pixels have similar normals: pair a pixel with itself as the pessimistic case
print('self pairs (n <= 32):', (n <= 32).mean())
# next-fit no-split packing of a stream (random order vs sorted-similar order)
def nextfit(seq):
    passes, fill = 0, 0
    for k in seq:
        if k == 0: continue
        if fill + k > 64:
            passes += 1; fill = 0
        fill += k
    return passes + (fill > 0)
print('A passes per pixel now (pixels with any live):', (n > 0).mean(), ' next-fit random order:', nextfit(n) / len(n), ' ideal split:', n.sum() / 64 / len(n))
ns = np.repeat(n[:5000], 4)   # runs of 4 identical pixels (coherent neighbours)
print('next-fit coherent order:', nextfit(ns) / len(ns))

def simulate(seq, R):
    """queue policy: push pixel's entries; run batches of 64 while >= 64; before reusing a ring slot whose pixel still has entries pending, drain."""
    from collections import deque
    q = deque()         # entries: pixel id
    passes = 0; lanes = 0
    pend = {}
    for p, k in enumerate(seq):
        old = p - R
        if old in pend and pend[old] > 0:
            # drain everything
            while q:
                c = min(64, len(q)); passes += 1; lanes += c
                for _ in range(c):
                    pend[q.popleft()] -= 1
        pend[p] = k
        q.extend([p] * k)
        while len(q) >= 64:
            passes += 1; lanes += 64
            for _ in range(64):
                pend[q.popleft()] -= 1
        pend.pop(old, None)
    while q:
        c = min(64, len(q)); passes += 1; lanes += c
        for _ in range(c): q.popleft()
    return passes / len(seq), lanes / max(passes, 1) / 64

for R in (2, 3, 4, 6, 8):
    print('R', R, 'random order: A passes/pixel %.3f fill %.2f' % simulate(list(n[:6000]), R), ' coherent: %.3f fill %.2f' % simulate(list(np.repeat(n[:1500], 4)), R))
