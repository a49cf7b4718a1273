"""renderutils operators (HIP) vs the outputs/gradients of the reference's own PyTorch module (tests/golden) and vs the
oracle at larger, broadcast and strided shapes.  Mirrors render/renderutils/tests/test_{bsdf,loss,mesh}.py, asserting."""
import pytest
import torch

from oracle import renderutils_ref as rr
from tests.util import load_npz, assert_close

pytestmark = pytest.mark.gpu


def _api():
    import nvdiffrecmc_amd.renderutils as ru
    return {
        'fresnel_shlick': ru._fresnel_shlick, 'ndf_ggx': ru._ndf_ggx, 'lambda_ggx': ru._lambda_ggx, 'masking_smith': ru._masking_smith,
        'lambert': ru.lambert, 'frostbite': ru.frostbite_diffuse, 'pbr_specular': ru.pbr_specular,
        'pbr_bsdf_lambert': lambda *a: ru.pbr_bsdf(*a, bsdf='lambert'), 'pbr_bsdf_frostbite': lambda *a: ru.pbr_bsdf(*a, bsdf='frostbite'),
        'prepare_shading_normal_11': lambda *a: ru.prepare_shading_normal(*a, two_sided_shading=True, opengl=True),
        'prepare_shading_normal_00': lambda *a: ru.prepare_shading_normal(*a, two_sided_shading=False, opengl=False),
        'prepare_shading_normal_bcast': lambda *a: ru.prepare_shading_normal(*a),
        'xfm_points': ru.xfm_points, 'xfm_vectors': ru.xfm_vectors, 'xfm_points_b': ru.xfm_points,
    }


NAMES = ['fresnel_shlick', 'ndf_ggx', 'lambda_ggx', 'masking_smith', 'lambert', 'frostbite', 'pbr_specular', 'pbr_bsdf_lambert',
         'pbr_bsdf_frostbite', 'prepare_shading_normal_11', 'prepare_shading_normal_00', 'prepare_shading_normal_bcast',
         'xfm_points', 'xfm_vectors', 'xfm_points_b']


@pytest.mark.parametrize('name', NAMES)
def test_op_vs_reference_module_vectors(name, dev):
    c = load_npz('renderutils_reference.npz')[name]
    ins = [torch.from_numpy(c['in%d' % i]).to(dev).requires_grad_(True) for i in range(int(c['n_in']))]
    out = _api()[name](*ins)
    assert_close(out.detach(), c['out'], 1e-5, floor=1e-4, what=name)           # reference tests print the same metric
    torch.nn.functional.mse_loss(out, torch.from_numpy(c['target']).to(dev)).backward()
    for i, t in enumerate(ins):
        if name.startswith('xfm') and i == 1:
            assert t.grad is None            # the plugin returns no matrix gradient (ops.py:514: (points_grad, None, ...))
            continue
        assert t.grad.shape == t.shape                                      # broadcast inputs get reduced gradients
        assert_close(t.grad, c['grad%d' % i], 2e-4, floor=1e-5, what='%s grad%d' % (name, i))


@pytest.mark.parametrize('loss,tm', [('l1', 'none'), ('l1', 'log_srgb'), ('mse', 'log_srgb'), ('smape', 'none'), ('relmse', 'none'),
                                     ('mse', 'none'), ('n2n', 'none')])
def test_image_loss_vs_reference_module_vectors(loss, tm, dev):
    import nvdiffrecmc_amd.renderutils as ru
    c = load_npz('renderutils_reference.npz')['image_loss_%s_%s' % (loss, tm)]
    ins = [torch.from_numpy(c['in%d' % i]).to(dev).requires_grad_(True) for i in range(2)]
    out = ru.image_loss(ins[0], ins[1], loss=loss, tonemapper=tm)
    out.backward()
    assert_close(out.detach(), c['out'], 1e-5)
    for i in range(2):
        assert_close(ins[i].grad, c['grad%d' % i], 1e-4, floor=1e-6)
    # a constant target (no gradient requested: the kernel then does not write one) leaves the image gradient unchanged
    img2 = ins[0].detach().clone().requires_grad_(True)
    ru.image_loss(img2, ins[1].detach(), loss=loss, tonemapper=tm).backward()
    assert torch.equal(img2.grad, ins[0].grad)


def test_image_loss_full_size_and_hdr_range(dev):
    """512x512 (more pixels than partial sums: grid-stride path), HDR values incl. out-of-range ones that must be clamped
    and get zero gradient (loss.cu:113-114,221-226)."""
    import nvdiffrecmc_amd.renderutils as ru
    g = torch.Generator().manual_seed(0)
    img = (torch.rand(2, 512, 512, 3, generator=g) * 3 - 0.5)
    img[0, 0, 0] = torch.tensor([70000.0, -2.0, 1.0])
    tgt = torch.rand(2, 512, 512, 3, generator=g) * 2
    for loss, tm in (('l1', 'log_srgb'), ('relmse', 'none'), ('mse', 'none')):
        a = img.clone().requires_grad_(True)
        ref = rr.image_loss(a, tgt, loss, tm, kernel_semantics=True)
        ref.backward()
        ad = img.to(dev).requires_grad_(True)
        out = ru.image_loss(ad, tgt.to(dev), loss=loss, tonemapper=tm)
        out.backward()
        assert_close(out.detach(), ref.detach(), 2e-5)
        inside = (img > 0) & (img < 65535)
        assert_close(ad.grad.cpu()[inside], a.grad[inside], 2e-4, floor=1e-9)
        assert ad.grad.cpu()[~inside].abs().max().item() == 0.0


def test_ops_broadcast_strided_and_errors(dev):
    import nvdiffrecmc_amd.renderutils as ru
    g = torch.Generator().manual_seed(2)
    R = lambda *s: torch.rand(*s, generator=g)
    # pbr_bsdf with broadcast view/light positions and a strided kd view
    big = R(2, 16, 12, 6)
    kd, arm = big[..., 0:3], big[..., 3:6]
    pos, nrm = R(2, 16, 12, 3), torch.nn.functional.normalize(R(2, 16, 12, 3), dim=-1)
    vp, lp = R(2, 1, 1, 3) + 2, R(1, 1, 1, 3) + 3
    ins = [kd, arm, pos, nrm, vp, lp]
    cpu = [t.clone().requires_grad_(True) for t in ins]
    ref = rr.pbr_bsdf(*cpu)
    ref.sum().backward()
    bigd = big.to(dev).requires_grad_(True)
    gpu = [bigd[..., 0:3], bigd[..., 3:6]] + [t.to(dev).requires_grad_(True) for t in ins[2:]]
    out = ru.pbr_bsdf(*gpu)
    out.sum().backward()
    assert_close(out.detach(), ref.detach(), 2e-4, floor=1e-3)   # GGX D near its peak is ill-conditioned in fp32
    assert_close(bigd.grad[..., 0:3], cpu[0].grad, 5e-4, floor=1e-3)
    assert gpu[4].grad.shape == (2, 1, 1, 3) and gpu[5].grad.shape == (1, 1, 1, 3)
    assert_close(gpu[5].grad, cpu[5].grad, 5e-4, floor=1e-3)
    # use_python switch exists and agrees
    assert_close(ru.lambert(gpu[3].detach(), gpu[2].detach(), use_python=True), ru.lambert(gpu[3].detach(), gpu[2].detach()), 1e-6)
    # argument errors (CHECK_TENSOR, torch_bindings.cpp:24-28)
    with pytest.raises(RuntimeError):
        ru.lambert(torch.rand(4, 4, 3, device=dev), torch.rand(4, 4, 3, device=dev))
    with pytest.raises(RuntimeError):
        ru.xfm_points(torch.rand(1, 10, 2, device=dev), torch.rand(1, 4, 4, device=dev))
    with pytest.raises(RuntimeError):
        ru.image_loss(torch.rand(1, 4, 4, 3, device=dev).double(), torch.rand(1, 4, 4, 3, device=dev).double())
    # anomaly mode finite check is kept (ops.py:107-108)
    with torch.autograd.set_detect_anomaly(True):
        with pytest.raises(AssertionError, match='inf or NaN'):
            ru._ndf_ggx(torch.full((1, 2, 2, 1), float('nan'), device=dev), torch.rand(1, 2, 2, 1, device=dev))


def test_light_update_pdf_vs_oracle(dev):
    from nvdiffrecmc_amd.light import EnvironmentLight
    from nvdiffrecmc_amd import scene as sc
    from oracle import oracle as orc
    for res in (256, 48):
        base = sc.env_map('E1', res)
        L = EnvironmentLight(base.to(dev))
        pdf, cols, rows = orc.light_update_pdf(base)
        assert_close(L._pdf, pdf, 1e-5, floor=1e-9)
        assert_close(L.cols, cols, 1e-5, floor=1e-6)
        assert_close(L.rows[:, 0], rows, 1e-5, floor=1e-6)
        assert L.rows.shape == (res, res)


def test_light_update_pdf_vs_reference_vectors(dev):
    """SURVEY a18 pin on the GPU: nvdr_light_update_pdf (through EnvironmentLight) against the tables the REFERENCE's own
    update_pdf produced (tests/golden/light_reference.npz, tools/make_golden.py gen_light)."""
    import numpy as np
    from nvdiffrecmc_amd.light import EnvironmentLight
    from tests.util import load_npz, checksum
    from tools import make_golden as mg
    gold_all = load_npz('light_reference.npz')
    for name, kind, res in mg.LIGHT_CASES:
        gold = gold_all[name]
        base = mg.light_case_base(kind, res, name)
        assert checksum(base) == str(gold['base_sha256'])
        L = EnvironmentLight(base.to(dev))
        assert_close(L._pdf, gold['pdf'], 1e-5, floor=1e-9, what=name + ' pdf')
        assert_close(L.cols, gold['cols'], 1e-5, floor=1e-6, what=name + ' cols')
        assert_close(L.rows, gold['rows'], 1e-5, floor=1e-6, what=name + ' rows')
        assert L.rows.shape == (res, res)


@pytest.mark.parametrize('cd,cs,bsdf', [(4, 4, 'pbr'), (3, 3, 'pbr'), (4, 3, 'diffuse')])
def test_shade_composite_vs_torch(cd, cs, bsdf, dev):
    """The fused final-colour op (render.py:119-127 + the division of optixutils/ops.py:139-141) against the torch
    expressions it replaces, forward and backward, with a broadcast ks ([1,1,1,3]) and a strided kd view."""
    import nvdiffrecmc_amd.renderutils as ru
    g = torch.Generator().manual_seed(5)
    R = lambda *s: torch.rand(*s, generator=g)
    N, H, W = 2, 19, 23
    diff, spec = R(N, H, W, cd) + 0.1, R(N, H, W, cs) + 0.1
    big, ks, go = R(N, H, W, 5), R(1, 1, 1, 3), R(N, H, W, 3)
    # the checker: the oracle's restatement of the reference lines, differentiated by torch autograd in double precision (and the
    # opt-in torch formulation of the package against it)
    from oracle import render_layer_ref as rl
    cpu = [t.double().requires_grad_(True) for t in (diff, spec, big, ks)]
    ref = rl.shade_composite(cpu[0], cpu[1], cpu[2][..., 1:4], cpu[3], bsdf)
    ref.backward(go.double())
    alt = ru.shade_composite(diff, spec, big[..., 1:4], ks, bsdf, use_python=True)
    assert_close(alt, ref.detach().float(), 1e-6)
    gpu = [t.to(dev).requires_grad_(True) for t in (diff, spec, big, ks)]
    out = ru.shade_composite(gpu[0], gpu[1], gpu[2][..., 1:4], gpu[3], bsdf)
    out.backward(go.to(dev))
    assert_close(out.detach(), ref.detach().float(), 1e-6)
    for a, b, name in zip(gpu, cpu, ('diff', 'spec', 'kd', 'ks')):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, name
            continue
        assert a.grad.shape == b.grad.shape, name
        assert_close(a.grad, b.grad.float(), 2e-5, floor=1e-5 * max(1.0, b.grad.abs().max().item()), what=name)
    with pytest.raises(RuntimeError):
        ru.shade_composite(torch.rand(1, 4, 4, 2, device=dev), gpu[1].detach(), gpu[2][..., 1:4].detach(), gpu[3].detach())


def test_fused_adam_matches_torch_adam_and_clamps(dev):
    """csrc/optim.hip (gradient scale + Adam + clamps in one launch, device-resident step counter) against torch.optim.Adam on
    the CPU followed by the reference's own sequence (train.py:439-476): lgt.base.grad *= 64, optimizer.step(), clamps."""
    from nvdiffrecmc_amd.optim import FusedAdam
    torch.manual_seed(5)
    shapes = [(300, 3), (3,), (17, 33, 3)]
    ks_min = torch.tensor([0.0, 0.08, 0.0])
    p_ref = [torch.nn.Parameter(torch.rand(*s) * 0.9 + 0.05) for s in shapes]
    p_gpu = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in p_ref]
    opt_ref = torch.optim.Adam(p_ref, lr=0.03)
    opt_gpu = FusedAdam(p_gpu, lr=0.03, grad_scales=[1.0, 1.0, 64.0],
                        clamps=[(0.0, 1.0), (None, 1.0, ks_min.to(dev)), (0.0, None)])
    for it in range(12):
        grads = [torch.randn(*s) * (0.3 if k < 2 else 0.3 / 64.0) for k, s in enumerate(shapes)]
        for p, q, g in zip(p_ref, p_gpu, grads):
            p.grad = g.clone()
            q.grad = g.clone().to(dev)
        p_ref[2].grad *= 64.0
        opt_ref.step()
        with torch.no_grad():
            p_ref[0].clamp_(0.0, 1.0)
            p_ref[1].copy_(torch.maximum(p_ref[1].clamp(max=1.0), ks_min))
            p_ref[2].clamp_(min=0.0)
        opt_gpu.step()
        for p, q in zip(p_ref, p_gpu):
            assert_close(q.detach().cpu(), p.detach(), 1e-6, floor=1.0)        # parameters are O(1): 1-3 ulp; one side of a clamp may sit at 0, the other at 1e-8
    assert opt_gpu.step_count == 12
    assert (p_gpu[0].min() >= 0) and (p_gpu[0].max() <= 1) and p_gpu[1][1] >= 0.08
    assert torch.equal(p_gpu[2].grad.cpu(), grads[2])           # the gradient itself is left alone (the scale is applied inside)


def test_fused_adam_material_set(dev):
    """The update of the reference's full parameter set in one launch (train.py:336-356,452-476): three learning rates (position /
    material / light), per-channel texture clamps from both sides (Texture2D.clamp_, ks_max = [0, 1, 1]), the normal map's
    clamp + normalize_, lgt.clamp_(min=0.01); a NaN parameter stays NaN; lr = 0 is a no-op on the parameters."""
    from nvdiffrecmc_amd.optim import FusedAdam
    torch.manual_seed(9)
    shapes = [(40, 40, 3), (40, 40, 3), (40, 40, 3), (16, 16, 3), (211, 3)]          # kd, ks, normal, light, v_pos
    ks_min, ks_max = torch.tensor([0.0, 0.08, 0.0]), torch.tensor([0.0, 1.0, 1.0])
    n_min, n_max = torch.tensor([-1.0, -1.0, 0.0]), torch.tensor([1.0, 1.0, 1.0])
    p_ref = [torch.nn.Parameter(torch.rand(*s) * 0.9 + 0.05) for s in shapes]
    p_gpu = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in p_ref]
    lr_mat, lr_lgt, lr_pos = 0.01, 0.03, 0.003
    opts = [torch.optim.Adam(p_ref[0:3], lr=lr_mat), torch.optim.Adam([p_ref[3]], lr=lr_lgt), torch.optim.Adam([p_ref[4]], lr=lr_pos)]
    opt_gpu = FusedAdam(p_gpu, lr=lr_mat, grad_scales=[1.0, 1.0, 1.0, 64.0, 1.0], lr_scales=[1.0, 1.0, 1.0, lr_lgt / lr_mat, lr_pos / lr_mat],
                        clamps=[(0.0, 1.0), (None, None, ks_min.to(dev), ks_max.to(dev)), (None, None, n_min.to(dev), n_max.to(dev)), (0.01, None), None],
                        normalize3=[False, False, True, False, False])
    for it in range(10):
        grads = [torch.randn(*s) * 0.3 for s in shapes]
        for p, q, g in zip(p_ref, p_gpu, grads):
            p.grad = g.clone()
            q.grad = g.clone().to(dev)
        p_ref[3].grad *= 64.0
        for o in opts:
            o.step()
        with torch.no_grad():
            p_ref[0].clamp_(0.0, 1.0)
            for i in range(3):
                p_ref[1][..., i].clamp_(min=float(ks_min[i]), max=float(ks_max[i]))
                p_ref[2][..., i].clamp_(min=float(n_min[i]), max=float(n_max[i]))
            p_ref[2].copy_(p_ref[2] / torch.sqrt(torch.clamp((p_ref[2] * p_ref[2]).sum(-1, keepdim=True), min=1e-20)))
            p_ref[3].clamp_(min=0.01)
        opt_gpu.step()
        for k, (p, q) in enumerate(zip(p_ref, p_gpu)):
            assert_close(q.detach().cpu(), p.detach(), 2e-6, floor=1.0, what='tensor %d step %d' % (k, it))
    assert float(p_gpu[1][..., 0].abs().max()) == 0.0                                   # ks.x is pinned to [0, 0]
    assert_close((p_gpu[2] * p_gpu[2]).sum(-1).cpu(), torch.ones(40, 40), 1e-6, floor=1.0)
    # NaN stays NaN (fmaxf / fminf would have turned it into a bound); lr = 0 leaves the parameters alone
    q = torch.nn.Parameter(torch.tensor([0.5, float('nan'), 0.25], device=dev))
    q.grad = torch.ones(3, device=dev)
    o2 = FusedAdam([q], lr=0.0, clamps=[(0.0, 1.0)])
    o2.step()
    assert torch.isnan(q[1]) and float(q[0]) == 0.5 and float(q[2]) == 0.25


def test_shading_frame_equals_the_composed_ops(dev):
    """The fused shading frame (shading normal, unit copy, shadow-ray origin in one launch) against prepare_shading_normal + the
    safe_normalize of the filter's guide + the offset of render.py:107; with an input that requires a gradient it composes them."""
    from nvdiffrecmc_amd import renderutils as ru
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 37, 53
    pos = torch.randn(N, H, W, 3, generator=g).to(dev)
    view = (torch.randn(N, 1, 1, 3, generator=g) * 3).to(dev)
    sn = torch.nn.functional.normalize(torch.randn(N, H, W, 3, generator=g), dim=-1).to(dev)
    st = torch.nn.functional.normalize(torch.randn(N, H, W, 3, generator=g), dim=-1).to(dev)
    gn = torch.nn.functional.normalize(sn.cpu() + 0.3 * torch.randn(N, H, W, 3, generator=g), dim=-1).to(dev)
    sn[0, :5] = 0.0                                                     # degenerate pixels (background)
    nrm, unit, ro = ru.shading_frame(pos, view, None, sn, st, gn, ro_eps=0.001)
    ref = ru.prepare_shading_normal(pos, view, None, sn, st, gn, two_sided_shading=True, opengl=True)
    assert torch.equal(nrm, ref)
    ref_unit = ref / torch.sqrt(torch.clamp(torch.sum(ref * ref, -1, keepdim=True), min=1e-20))
    assert torch.equal(unit, ref_unit)          # same association as torch.sum: the filter raises dot products of it to the 128th power
    assert torch.equal(ro, pos + ref * 0.001)
    # differentiable through the shading normal (one launch: prepare_shading_normal_bwd), for every input incl. a per-pixel perturbed
    # normal and the broadcast view position; the unit copy and the ray origin carry no gradient (the reference's graph has none there)
    pn = torch.nn.functional.normalize(torch.randn(N, H, W, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
    up = torch.randn(N, H, W, 3, generator=g).to(dev)
    a = [t.clone().requires_grad_(True) for t in (pos, view, pn, sn, st, gn)]
    b = [t.clone().requires_grad_(True) for t in (pos, view, pn, sn, st, gn)]
    n2, u2, r2 = ru.shading_frame(*a)
    assert n2.requires_grad and not u2.requires_grad and not r2.requires_grad
    n3 = ru.prepare_shading_normal(*b, two_sided_shading=True, opengl=True)
    assert torch.equal(n2.detach(), n3.detach())
    (n2 * up).sum().backward()
    (n3 * up).sum().backward()
    for x, y in zip(a, b):
        assert x.grad.shape == y.grad.shape
        if x.shape == x.grad.shape and x.shape[1] > 1:
            assert torch.equal(x.grad, y.grad)
        else:       # the broadcast view position: its full-extent gradient is folded over H and W, in another order than autograd folds it
            assert_close(x.grad, y.grad, 1e-5, floor=float(y.grad.abs().max()))


def test_lookup_rows_matches_torch_indexing(dev):
    """csrc/renderutils.hip gather_rows (the texture lookup of the iteration harness in one launch) against torch indexing: rows by
    index with -1 = zero row, and its adjoint (an index_add)."""
    from nvdiffrecmc_amd.trainer import _lookup_rows
    g = torch.Generator().manual_seed(4)
    tex = torch.rand(500, 3, generator=g).to(dev).requires_grad_(True)
    idx = torch.randint(-1, 500, (4000,), generator=g).to(torch.int32).to(dev)
    out = _lookup_rows.apply(tex, idx)
    ref = torch.where((idx >= 0)[:, None], tex.detach()[idx.clamp(min=0).long()], torch.zeros(1, 3, device=dev))
    assert torch.equal(out.detach(), ref)
    w = torch.rand(4000, 3, generator=g).to(dev)
    out.backward(w)
    gref = torch.zeros(500, 3, device=dev).index_add_(0, idx.clamp(min=0).long(), w * (idx >= 0)[:, None])
    assert_close(tex.grad, gref, 1e-6, floor=1e-3)


@pytest.mark.parametrize('channels', [4, 3], ids=['filter_output_rgbw', 'plain_rgb'])
@pytest.mark.parametrize('loss,tonemapper', [('l1', 'log_srgb'), ('mse', 'none'), ('smape', 'log_srgb')])
def test_fused_composite_and_mean_loss_equals_the_separate_kernels(loss, tonemapper, channels, dev):
    """shade_composite_loss (round 6: composite + mean image loss + both adjoints in ONE launch) against shade_composite -> image_loss_mean and their
    autograd adjoints: the loss and the four gradients bit for bit, for the (colour sum, weight) output of the filter kernel and for plain RGB,
    with out-of-range pixels (the loss clamps in forward and zeroes the gradient there) and another upstream gradient than the one it was told."""
    import nvdiffrecmc_amd.renderutils as ru
    g = torch.Generator().manual_seed(31)
    N, H, W = 2, 37, 53

    def R(*s, lo=0.0, hi=1.0):
        return (torch.rand(*s, generator=g) * (hi - lo) + lo).to(dev)
    diff, spec = R(N, H, W, channels, lo=-0.2, hi=3.0), R(N, H, W, channels, lo=-0.2, hi=3.0)
    if channels == 4:
        diff[..., 3], spec[..., 3] = R(N, H, W, lo=0.3, hi=2.0), R(N, H, W, lo=0.3, hi=2.0)
    diff[0, 0, 0, 0] = 7.0e4                                  # beyond the loss's clamp
    kd, ks, target = R(N, H, W, 3), R(N, H, W, 3), R(N, H, W, 3, hi=2.0)
    one = torch.ones((), device=dev)
    leaves_a = [t.clone().requires_grad_(True) for t in (diff, spec, kd, ks)]
    la = ru.image_loss_mean(ru.shade_composite(*leaves_a), target, loss=loss, tonemapper=tonemapper)
    la.backward(gradient=one)
    leaves_b = [t.clone().requires_grad_(True) for t in (diff, spec, kd, ks)]
    lb = ru.shade_composite_loss(*leaves_b, target, one, loss=loss, tonemapper=tonemapper)
    lb.backward(gradient=one)
    assert torch.equal(la, lb)
    for a, b, name in zip(leaves_a, leaves_b, ('diff', 'spec', 'kd', 'ks')):
        assert torch.equal(a.grad, b.grad), name
    # an upstream gradient other than the announced one: rescaled in backward
    leaves_c = [t.clone().requires_grad_(True) for t in (diff, spec, kd, ks)]
    lc = ru.shade_composite_loss(*leaves_c, target, one, loss=loss, tonemapper=tonemapper)
    lc.backward(gradient=torch.full((), 2.5, device=dev))
    for a, c in zip(leaves_a, leaves_c):
        assert torch.allclose(2.5 * a.grad, c.grad, rtol=1e-6, atol=0.0)
