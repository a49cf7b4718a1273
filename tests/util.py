"""Shared helpers for the parity tests."""
import hashlib
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_npz(name):
    d = np.load(os.path.join(ROOT, 'tests', 'golden', name), allow_pickle=False)
    out = {}
    for k in d.files:
        head, _, tail = k.rpartition('/')
        out.setdefault(head, {})[tail] = d[k]
    return out


def checksum(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def rel_err(got, ref, floor=1e-3):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs() / (ref.abs() + floor))


def assert_close(got, ref, rtol, floor=1e-3, frac_outliers=0.0, what=''):
    """|got - ref| <= rtol * (|ref| + floor) everywhere, except for at most `frac_outliers` of the elements."""
    r = rel_err(got, ref, floor)
    bad = int((r > rtol).sum())
    allowed = int(frac_outliers * r.numel())
    assert bad <= allowed, '%s: %d of %d elements exceed rtol %.1e (max %.3e, allowed outliers %d)' % (
        what, bad, r.numel(), rtol, r.max().item(), allowed)


# ---- the per-pixel RNG of the raygen program (kernel.cu:30-45,504-505) in numpy: which permutation rows a pixel uses

def _rand_pcg(state):
    """(output, advanced state) of rand_pcg on uint32 arrays (wrap-around arithmetic)."""
    s = state.astype(np.uint32)
    with np.errstate(over='ignore'):
        word = ((s >> ((s >> np.uint32(28)) + np.uint32(4))) ^ s) * np.uint32(277803737)
        nxt = s * np.uint32(747796405) + np.uint32(2891336453)
    return (word >> np.uint32(22)) ^ word, nxt


def perm_rows(lin, seed, n_perms, pixel_index_offset=0):
    """(lightIdx, bsdfIdx) of the pixels with linear indices `lin` (numpy int array): kernel.cu:504-505."""
    a = np.full(lin.shape, seed, dtype=np.uint32)
    b = (lin.astype(np.int64) + pixel_index_offset).astype(np.uint32)
    rng = _rand_pcg(a)[0] ^ _rand_pcg(b)[0]
    li, rng = _rand_pcg(rng)
    bi, rng = _rand_pcg(rng)
    return (li % np.uint32(n_perms)).astype(np.int64), (bi % np.uint32(n_perms)).astype(np.int64)


def vis_by_sample_from_stratum_bits(bits, perms, seed, S, pixel_index_offset=0):
    """Convert the GPU's visibility planes (int32 [P, 2, ceil(S/32)]: bit s = the ray of STRATUM s is occluded; csrc/env_shade.hip
    stage 3) into the oracle's vis_in layout (uint8 [P, 2S]: entry 2i / 2i+1 = light / BSDF sample i is UNoccluded), through
    the permutation rows the pixel's RNG picks (sample i of a pixel uses stratum perms[row][i])."""
    bits = np.ascontiguousarray(bits).view(np.uint32)
    P = bits.shape[0]
    lin = np.arange(P)
    li, bi = perm_rows(lin, seed, perms.shape[0], pixel_index_offset)
    out = np.ones((P, 2 * S), dtype=np.uint8)
    for plane, rows in ((0, li), (1, bi)):
        strat = perms[rows].astype(np.int64)                          # [P, S]: stratum of sample i
        w = np.take_along_axis(bits[:, plane, :], strat >> 5, axis=1)
        occ = (w >> (strat & 31).astype(np.uint32)) & np.uint32(1)
        out[:, plane::2] = (1 - occ).astype(np.uint8)
    return out
