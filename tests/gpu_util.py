"""Helpers shared by the -m gpu tests: upload oracle scene inputs, call the public API."""
import torch

from oracle import scene_cpu


def to_dev(kw, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}


def make_ctx(mesh, dev):
    from nvdiffrecmc_amd import optixutils as ou
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, mesh['v_pos'].to(dev), mesh['t_pos_idx'].to(dev), rebuild=1)
    return ctx


def gpu_env_shade(ctx, kw, dev, bsdf, n, seed, dg=None, sg=None, cache_vis=False, shadow_scale=1.0, offset=0):
    """Forward (and backward when dg/sg given) through nvdiffrecmc_amd.optixutils.optix_env_shade."""
    from nvdiffrecmc_amd import optixutils as ou
    ou.ops.set_permutation_table(n, kw['perms'].to(dev))
    ou.ops.set_pixel_index_offset(offset)
    ou.ops._optix_env_shade_func.cache_visibility = cache_vis
    g = {k: v.to(dev) for k, v in kw.items() if k != 'perms'}
    leaves = ('gb_pos', 'gb_normal', 'gb_kd', 'gb_ks', 'light')
    if dg is not None:
        for k in leaves:
            g[k] = g[k].clone().requires_grad_(True)
    diff, spec = ou.optix_env_shade(ctx, g['mask'], g['ro'], g['gb_pos'], g['gb_normal'], g['gb_view_pos'], g['gb_kd'], g['gb_ks'],
                                    g['light'], g['pdf'], g['rows'], g['cols'], BSDF=bsdf, n_samples_x=n, rnd_seed=seed,
                                    shadow_scale=shadow_scale)
    out = {'diff': diff.detach().cpu(), 'spec': spec.detach().cpu()}
    if dg is not None:
        ((diff * dg.to(dev)).sum() + (spec * sg.to(dev)).sum()).backward()
        out.update({'gb_pos_grad': g['gb_pos'].grad.cpu(), 'gb_normal_grad': g['gb_normal'].grad.cpu(),
                    'gb_kd_grad': g['gb_kd'].grad.cpu(), 'gb_ks_grad': g['gb_ks'].grad.cpu(), 'light_grad': g['light'].grad.cpu()})
    ou.ops.set_pixel_index_offset(0)
    return out
