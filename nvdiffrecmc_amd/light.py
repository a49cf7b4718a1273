"""EnvironmentLight: the producer of the light / pdf / rows / cols inputs of optix_env_shade.

Mirrors the parts of render/light.py:21-59 that are on the hot path (the HDR load/save helpers need
imageio + nvdiffrast and are out of scope).  update_pdf() runs as three HIP launches
(csrc/light.hip) instead of the reference's ~10 small torch kernels per training iteration.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class EnvironmentLight:
    LIGHT_MIN_RES = 16
    MIN_ROUGHNESS = 0.08
    MAX_ROUGHNESS = 0.5

    def __init__(self, base):
        self.mtx = None
        self.base = base
        self.pdf_scale = (self.base.shape[0] * self.base.shape[1]) / (2 * np.pi * np.pi)
        self.update_pdf()

    def xfm(self, mtx):
        self.mtx = mtx

    def parameters(self):
        return [self.base]

    def clone(self):
        return EnvironmentLight(self.base.clone().detach())

    def clamp_(self, min=None, max=None):
        self.base.clamp_(min, max)

    def update_pdf(self):
        with torch.no_grad():
            base = self.base.detach()
            _lib.require_cuda_f32(base, 'EnvironmentLight.base')
            base = base.contiguous()
            H, W = base.shape[0], base.shape[1]
            self._pdf = torch.empty(H, W, dtype=torch.float32, device=base.device)
            self.cols = torch.empty(H, W, dtype=torch.float32, device=base.device)
            rows = torch.empty(H, dtype=torch.float32, device=base.device)
            _lib.check(_lib.load().nvdr_light_update_pdf(_lib.ptr(base), H, W, _lib.ptr(self._pdf), _lib.ptr(self.cols),
                                                         _lib.ptr(rows), _lib.stream_ptr()), 'light_update_pdf')
            # the reference materialises rows as [H,W] with identical columns and passes rows[:,0]
            # (render.py:114); keep that shape so callers written against it index the same way
            self.rows = rows[:, None].expand(H, W)


def create_trainable_env_rnd(base_res, scale=0.5, bias=0.25, device='cuda'):
    base = torch.rand(base_res, base_res, 3, dtype=torch.float32, device=device) * scale + bias
    return EnvironmentLight(base.clone().detach().requires_grad_(True))
