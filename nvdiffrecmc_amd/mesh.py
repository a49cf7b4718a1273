"""Vertex normals / tangents of a trained mesh and their adjoint on MI355X (csrc/mesh.hip).

The reference computes them per iteration with torch scatter_add_ on the GPU (render/mesh.py:150-219, called from
geometry/dlmesh.py:52-54) and lets autograd differentiate them.  Here both directions are gathers over a vertex -> corner
adjacency that is built once per topology: a fixed summation order (bit-reproducible vertex frames; atomics are not), one launch
forward and two backward instead of ~40 small torch kernels.
"""
import ctypes

import torch

from . import _lib


class MeshTopology:
    """Index buffers of a mesh + the vertex -> (triangle, corner) adjacency nvdr_mesh_args asks for.  Normals and tangents are
    indexed like positions (mesh.py:178,219: t_nrm_idx = t_tng_idx = t_pos_idx)."""

    def __init__(self, t_pos_idx, n_verts, v_tex=None, t_tex_idx=None):
        _lib.require_cuda_f32(t_pos_idx, 't_pos_idx', torch.int32)
        self.t_pos_idx = t_pos_idx.contiguous()
        self.n_verts, self.n_tris = int(n_verts), int(t_pos_idx.shape[0])
        self.v_tex = self.t_tex_idx = None
        if v_tex is not None:
            _lib.require_cuda_f32(v_tex, 'v_tex')
            _lib.require_cuda_f32(t_tex_idx, 't_tex_idx', torch.int32)
            if t_tex_idx.shape[0] != self.n_tris:
                raise RuntimeError('t_tex_idx must have one row per triangle')
            self.v_tex, self.t_tex_idx = v_tex.contiguous(), t_tex_idx.contiguous()
        flat = self.t_pos_idx.reshape(-1).long()
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= self.n_verts):
            raise RuntimeError('t_pos_idx refers to vertices outside [0, %d)' % self.n_verts)
        order = torch.sort(flat, stable=True)[1]                # (triangle, corner) ascending inside a vertex
        self.adj_corner = order.to(torch.int32).contiguous()
        counts = torch.bincount(flat, minlength=self.n_verts)
        self.adj_start = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32).contiguous()

    def args(self, v_pos):
        a = _lib.NvdrMeshArgs()
        a.v_pos, a.n_verts = v_pos.data_ptr(), self.n_verts
        a.t_pos_idx, a.n_tris = self.t_pos_idx.data_ptr(), self.n_tris
        a.v_tex = self.v_tex.data_ptr() if self.v_tex is not None else None
        a.t_tex_idx = self.t_tex_idx.data_ptr() if self.t_tex_idx is not None else None
        a.adj_start, a.adj_corner = self.adj_start.data_ptr(), self.adj_corner.data_ptr()
        return a


class _mesh_frame_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_pos, topo):
        _lib.require_cuda_f32(v_pos, 'v_pos')
        if v_pos.dim() != 2 or v_pos.shape[0] != topo.n_verts or v_pos.shape[1] != 3:
            raise RuntimeError('v_pos must be [%d, 3] (got %s)' % (topo.n_verts, tuple(v_pos.shape)))
        v_pos = v_pos.contiguous()
        v_nrm = torch.empty_like(v_pos)
        v_tng = torch.empty_like(v_pos) if topo.v_tex is not None else None
        a = topo.args(v_pos)
        _lib.check(_lib.load().nvdr_mesh_frame_fwd(ctypes.byref(a), _lib.ptr(v_nrm), _lib.ptr(v_tng), _lib.stream_ptr()), 'mesh_frame_fwd')
        ctx.save_for_backward(v_pos)
        ctx.topo = topo
        if v_tng is None:
            v_tng = torch.zeros_like(v_pos)
            ctx.mark_non_differentiable(v_tng)
        return v_nrm, v_tng

    @staticmethod
    def backward(ctx, g_nrm, g_tng):
        v_pos, = ctx.saved_tensors
        topo = ctx.topo
        g_nrm = g_nrm.contiguous() if g_nrm is not None else None
        g_tng = g_tng.contiguous() if (g_tng is not None and topo.v_tex is not None) else None
        scratch = torch.empty(topo.n_verts, 6, dtype=torch.float32, device=v_pos.device)
        g_pos = torch.empty_like(v_pos)
        a = topo.args(v_pos)
        _lib.check(_lib.load().nvdr_mesh_frame_bwd(ctypes.byref(a), _lib.ptr(g_nrm), _lib.ptr(g_tng), _lib.ptr(scratch), _lib.ptr(g_pos), 0,
                                                   _lib.stream_ptr()), 'mesh_frame_bwd')
        return g_pos, None


def mesh_frame(v_pos, topo):
    """(v_nrm, v_tng) = (auto_normals, compute_tangents) of the mesh (v_pos, topo); differentiable w.r.t. v_pos."""
    return _mesh_frame_func.apply(v_pos, topo)
