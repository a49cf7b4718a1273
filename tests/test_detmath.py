"""include/nvdr_detmath.h: accuracy against double-precision libm (CPU) and bit-equality device vs host (GPU)."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc


def _inputs(op, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    if op in ('sin', 'cos'):
        x = (torch.rand(n, generator=g) * 2 - 1) * (4 * math.pi)
        return x, x
    if op == 'acos':
        x = torch.rand(n, generator=g) * 2 - 1
        x[:6] = torch.tensor([-1.0, 1.0, 0.5, -0.5, 0.0, 0.99999994])
        return x, x
    x, y = torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 2 - 1
    x[:4] = torch.tensor([0.0, 0.0, 1.0, -1.0])
    y[:4] = torch.tensor([0.0, -1.0, 0.0, 0.0])
    return x, y


@pytest.mark.parametrize('op', ['sin', 'cos', 'acos', 'atan2'])
def test_detmath_accuracy_vs_libm(op):
    x, y = _inputs(op, 1 << 20)
    got = orc.detmath(op, x, y).double().numpy()
    xd, yd = x.double().numpy(), y.double().numpy()
    ref = {'sin': np.sin, 'cos': np.cos, 'acos': np.arccos}[op](xd) if op != 'atan2' else np.arctan2(xd, yd)
    err = np.abs(got - ref)
    if op in ('sin', 'cos'):
        assert err.max() < 1.5e-7          # ~1.5 ulp at 1.0, absolute near the zeros
    else:
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert (err / ulp).max() <= 3.5


@pytest.mark.gpu
@pytest.mark.parametrize('op', ['sin', 'cos', 'acos', 'atan2'])
def test_detmath_device_equals_host_bitwise(op, dev):
    import ctypes
    from nvdiffrecmc_amd import _lib
    lib = _lib.load()
    x, y = _inputs(op, 1 << 21, seed=1)
    host = orc.detmath(op, x, y)
    xd, yd = x.to(dev), y.to(dev)
    out = torch.empty_like(xd)
    _lib.check(lib.nvdr_test_detmath(['sin', 'cos', 'acos', 'atan2'].index(op), _lib.ptr(xd), _lib.ptr(yd), x.numel(),
                                     _lib.ptr(out), _lib.stream_ptr()), 'detmath')
    got = out.cpu()
    assert torch.equal(got.view(torch.int32), host.view(torch.int32)), \
        '%d mismatching bit patterns' % int((got.view(torch.int32) != host.view(torch.int32)).sum())
