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
