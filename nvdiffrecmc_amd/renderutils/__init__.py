# Same public surface as render/renderutils/__init__.py:9-10 of the reference.
from .ops import (xfm_points, xfm_vectors, image_loss, prepare_shading_normal, lambert, frostbite_diffuse,
                  pbr_specular, pbr_bsdf, _fresnel_shlick, _ndf_ggx, _lambda_ggx, _masking_smith, shade_composite, shading_frame, image_loss_mean, shade_composite_loss)
# shade_composite, shading_frame, image_loss_mean and shade_composite_loss are additive (the reference composes the final colour in torch, render/render.py:119-127): importable,
# but __all__ stays the reference's list
__all__ = ["xfm_vectors", "xfm_points", "image_loss", "prepare_shading_normal", "lambert", "frostbite_diffuse",
           "pbr_specular", "pbr_bsdf", "_fresnel_shlick", "_ndf_ggx", "_lambda_ggx", "_masking_smith"]
