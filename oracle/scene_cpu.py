"""CPU-side scene assembly for the oracle (test infrastructure): primary hits by brute force
(oracle.closest), then the same torch G-buffer interpolation the GPU harness uses."""
import torch

from nvdiffrecmc_amd import scene as sc
from . import oracle as orc


def make_inputs(mesh_name='bob', H=128, W=128, n_samples_x=2, view=0, n_views=8, env='E1', probe_res=256,
                n_threads=None, kd_mode='texture', perms_seed=0):
    """Everything optix_env_shade needs (render/render.py:99-116), on the CPU, for one view."""
    nt = n_threads or orc.max_threads()
    mesh = sc.load_mesh(mesh_name)
    mv, mvp, campos = sc.camera(view, n_views, aspect=W / H)
    ro, rd = sc.primary_rays(mv, H, W)
    t, tri, uv = orc.closest(mesh['v_pos'], mesh['t_pos_idx'], ro.reshape(-1, 3), rd.reshape(-1, 3), n_threads=nt)
    gb = sc.gbuffer_from_hits(mesh, t.reshape(H, W), tri.reshape(H, W), uv.reshape(H, W, 2), ro, rd, kd_mode=kd_mode)
    base = sc.env_map(env, probe_res)
    pdf, rows, cols = sc.light_tables(base)
    inp = {
        'mesh': mesh,
        'mask': gb['mask'],
        'gb_pos': gb['gb_pos'],
        'gb_normal': gb['gb_normal'],
        'gb_view_pos': campos[None, None, None, :].contiguous(),
        'gb_kd': gb['kd'],
        'gb_ks': gb['ks'],
        'depth': gb['depth'],
        'light': base,
        'pdf': pdf,
        'rows': rows[:, 0],          # strided column view, exactly what render.py:114 passes
        'cols': cols,
        'perms': sc.perms_table(n_samples_x, seed=perms_seed),
        'n_samples_x': n_samples_x,
    }
    inp['ro'] = (inp['gb_pos'] + inp['gb_normal'] * 0.001).contiguous()  # render.py:110
    return inp


def shade_kwargs(inp):
    keys = ('mask', 'ro', 'gb_pos', 'gb_normal', 'gb_view_pos', 'gb_kd', 'gb_ks', 'light', 'pdf', 'rows', 'cols', 'perms')
    return {k: inp[k] for k in keys}
