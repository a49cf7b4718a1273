"""BilateralDenoiser module, same interface as denoiser/denoiser.py:17-31 of the reference."""
import math

import torch

from . import optixutils as ou


def _safe_normalize(x, eps=1e-20):
    # render/util.py: x / sqrt(clamp(dot(x, x), min=eps))
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


class BilateralDenoiser(torch.nn.Module):
    def __init__(self, influence=1.0):
        super(BilateralDenoiser, self).__init__()
        self.set_influence(influence)

    def set_influence(self, factor):
        self.sigma = max(factor * 2, 0.0001)
        self.variance = self.sigma ** 2.
        self.N = 2 * math.ceil(self.sigma * 2.5) + 1

    def forward(self, input):
        col = input[..., 0:3]
        nrm = _safe_normalize(input[..., 3:6])  # bent normals can be shorter than 1
        zdz = input[..., 6:8]
        return ou.bilateral_denoiser(col, nrm, zdz, self.sigma)
