# Same public surface as render/optixutils/__init__.py:9-10 of the reference, plus additive ray-query helpers.
from . import ops
from .ops import OptiXContext, optix_build_bvh, optix_env_shade, bilateral_denoiser
from .ops import trace_visibility, trace_visibility_wide, trace_closest, render_gbuffer  # additive (test hooks / G-buffer producer)
__all__ = ["OptiXContext", "optix_build_bvh", "optix_env_shade", "bilateral_denoiser"]
