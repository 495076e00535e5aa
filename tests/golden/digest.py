"""Compact fingerprint of a set of named tensors (parameter gradients): per tensor its largest magnitude, its L2 norm, four
projections on seeded random directions and 32 strided elements -- a few hundred KB pin gradients whose full size would be 30 MB.
Shared by tests/golden/make_golden.py (the reference's gradients) and the GPU tests (the engine's)."""
import zlib

import numpy as np


def tensor_digest(name, a):
    a = np.asarray(a, dtype=np.float64).ravel()
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    proj = [float(np.dot(a, rs.standard_normal(a.size))) for _ in range(4)]
    step = max(1, a.size // 32)
    return np.array([float(np.abs(a).max()), float(np.sqrt(np.dot(a, a)))] + proj + a[::step][:32].tolist() + [0.0] * (32 - len(a[::step][:32])), dtype=np.float64)


def digest(named):
    return {k: tensor_digest(k, v) for k, v in named.items()}


def compare(ref, got, rtol):
    """max over tensors of |digest difference| / (largest |gradient element| of the reference tensor, or its projection scale)"""
    worst = (0.0, None)
    for k, r in ref.items():
        g = got[k]
        amax, norm = r[0], r[1]
        if amax == 0.0:
            assert g[0] == 0.0, k
            continue
        dev = max(abs(g[0] - r[0]) / amax, abs(g[1] - r[1]) / norm,
                  float(np.abs(g[2:6] - r[2:6]).max()) / norm,          # projections on unit-variance directions scale with the norm
                  float(np.abs(g[6:] - r[6:]).max()) / amax)
        if dev > worst[0]:
            worst = (dev, k)
    assert worst[0] <= rtol, "gradient digest deviates %.3g (> %.3g) at %s" % (worst[0], rtol, worst[1])
    return worst


def logit_rows(out_len, per_clip=8):
    """The output positions of every clip whose logits a large-size fixture keeps (config 5: 8 of 10 885 rows per clip; the first and last included)."""
    return np.unique(np.linspace(0, out_len - 1, per_clip).astype(np.int64))
