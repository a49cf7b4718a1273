"""csrc/ieee_arith.h: the division / square root of the shading kernels without the compiler's range-scaling steps must round exactly
like the compiler's own `/`, sqrtf and sqrt inside the documented domain (bsdf.h / kernel.cu of the reference fix the rounding: IEEE)."""
import pytest
import torch


@pytest.mark.gpu
def test_unscaled_division_and_square_root_round_like_the_plain_operators(dev):
    from nvdiffrecmc_amd import _lib
    lib = _lib.load()
    c = torch.zeros(7, dtype=torch.int64, device=dev)
    _lib.check(lib.nvdr_test_arith(_lib.ptr(c), _lib.stream_ptr()), 'arith')
    torch.cuda.synchronize()
    sqrt_bad, sqrt_below, div_bad, special_bad, ddiv_bad, dsqrt_bad, n = [int(v) for v in c.cpu()]
    print('comparisons %d; sqrt mismatches among the positive floats below 2^-96 and the negative denormals (outside the domain): %d' % (n, sqrt_below))
    assert n == (1 << 32) + 3 * (1 << 30) + 13 * 9 + 3 * (1 << 28) + 2 * 0x3f800000 + 5
    assert sqrt_bad == 0, 'nvdr_sqrt differs from sqrtf on %d floats of its domain' % sqrt_bad
    assert div_bad == 0, 'nvdr_div differs from `/` on %d pairs of its domain' % div_bad
    assert special_bad == 0, 'nvdr_div differs from `/` on %d special-value pairs' % special_bad
    assert ddiv_bad == 0, 'nvdr_ddiv differs from `/` on %d double pairs' % ddiv_bad
    assert dsqrt_bad == 0, 'nvdr_dsqrt differs from sqrt on %d doubles' % dsqrt_bad
