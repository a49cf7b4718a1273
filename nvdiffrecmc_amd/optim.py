"""Adam + gradient scale + parameter clamps of one iteration in ONE kernel launch (csrc/optim.hip).

Same update as torch.optim.Adam(params, lr, betas, eps) without weight decay / amsgrad (the optimizer of the reference,
train.py:452-461), followed by the clamps the reference applies to its parameters after the step (train.py:470-476); the
multiplication of the light gradient by 64 (train.py:439-440) enters as a per-tensor gradient scale.  The step counter lives in
device memory, so the launch can be captured in a HIP graph.
"""
import math

import torch

from . import _lib


class FusedAdam:
    """params: list of tensors (requires_grad, contiguous fp32, on one GPU).
    clamps[i]: None or (lo, hi[, lo_vec[, hi_vec]]) with lo / hi floats or None and lo_vec / hi_vec optional per-channel bounds
    (device tensors: element e is bounded by lo_vec[e % len(lo_vec)] / hi_vec[e % len(hi_vec)]; Texture2D.clamp_).
    grad_scales[i]: factor applied to the gradient of parameter i inside the update (p.grad itself is left alone).
    lr_scales[i]: learning rate of parameter i relative to lr (the reference runs three Adams with position / material / light
    rates, train.py:336-356).  normalize3[i]: renormalise every texel of three channels after the clamps (the normal map).
    sparse[i]: the tensor is a three-channel texture whose gradient is zero almost everywhere (a nearest-texel lookup): tiles of 64
    texels without gradient and without history are skipped -- same parameters, a fraction of the traffic (csrc/optim.hip).
    zero_grad[i] (with sparse): p.grad is zeroed behind the update (a persistent scatter-add buffer then needs no memset)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clamps=None, grad_scales=None, lr_scales=None, normalize3=None,
                 sparse=None, zero_grad=None):
        self.params = list(params)
        if not 1 <= len(self.params) <= 8:
            raise ValueError('FusedAdam: 1..8 parameter tensors')
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise ValueError('FusedAdam: parameters must be contiguous fp32 tensors on the GPU')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.clamps = list(clamps) if clamps is not None else [None] * len(self.params)
        self.grad_scales = list(grad_scales) if grad_scales is not None else [1.0] * len(self.params)
        self.lr_scales = list(lr_scales) if lr_scales is not None else [1.0] * len(self.params)
        self.normalize3 = list(normalize3) if normalize3 is not None else [False] * len(self.params)
        self.sparse = list(sparse) if sparse is not None else [False] * len(self.params)
        self.zero_grad_after = list(zero_grad) if zero_grad is not None else [False] * len(self.params)
        self.active = [torch.zeros((p.numel() // 3 + 63) // 64, dtype=torch.uint8, device=p.device) if sp else None for p, sp in zip(self.params, self.sparse)]
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(8, dtype=torch.int32, device=self.params[0].device)     # steps taken, scratch, beta1^t, beta2^t (doubles)

    @property
    def step_count(self):
        return int(self.state[0].item())

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self, subset=None, advance=True, grad_mult=1.0):
        """subset: indices of the tensors to update in this launch (None = all); advance=False leaves the step counter alone (every
        launch of an iteration but the last, nvdr_adam_step_partial); grad_mult: extra factor on the gradients of this launch (1 / world
        after a summing all-reduce)."""
        idx = list(range(len(self.params))) if subset is None else list(subset)
        tab = (_lib.NvdrAdamTensor * len(idx))()
        keep = []
        for j, i in enumerate(idx):
            p = self.params[i]
            if p.grad is None:
                raise RuntimeError('FusedAdam.step: parameter %d has no gradient' % i)
            if self.zero_grad_after[i] and not p.grad.is_contiguous():
                # a copy would be zeroed instead of the buffer the producer scatter-adds into
                raise RuntimeError('FusedAdam.step: zero_grad needs a contiguous .grad for parameter %d (the persistent scatter-add buffer itself)' % i)
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            keep.append(g)
            lo, hi, lo_vec, hi_vec = -math.inf, math.inf, None, None
            if self.clamps[i] is not None:
                c = self.clamps[i]
                lo = -math.inf if c[0] is None else float(c[0])
                hi = math.inf if c[1] is None else float(c[1])
                lo_vec = c[2] if len(c) > 2 else None
                hi_vec = c[3] if len(c) > 3 else None
            t = tab[j]
            t.param, t.grad, t.exp_avg, t.exp_avg_sq = p.data_ptr(), g.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            t.n = p.numel()
            t.grad_scale, t.lo, t.hi = float(self.grad_scales[i]) * float(grad_mult), lo, hi
            t.lo_vec = lo_vec.data_ptr() if lo_vec is not None else None
            t.lo_vec_n = lo_vec.numel() if lo_vec is not None else 0
            t.hi_vec = hi_vec.data_ptr() if hi_vec is not None else None
            t.hi_vec_n = hi_vec.numel() if hi_vec is not None else 0
            # lr_scales[i] == 0 freezes the tensor (the C block's lr_scale 0 means "the plain rate": a zero-initialised block trains)
            t.lr_scale, t.normalize3 = float(self.lr_scales[i]) or 1.0, int(bool(self.normalize3[i]))
            t.frozen = int(float(self.lr_scales[i]) == 0.0)
            t.active = self.active[i].data_ptr() if self.active[i] is not None else None
            t.zero_grad = int(bool(self.zero_grad_after[i] and self.active[i] is not None))
        with torch.no_grad():
            _lib.check(_lib.load().nvdr_adam_step_partial(tab, len(idx), self.lr, self.betas[0], self.betas[1], self.eps,
                                                          _lib.ptr(self.state), int(bool(advance)), _lib.stream_ptr()), 'adam_step')
