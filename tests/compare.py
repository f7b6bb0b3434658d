"""Comparison helpers shared by the tests, and the compact form of a large reference tensor.

A fixture that would hold a large float tensor of the reference (a block's output, an input gradient, a wide weight gradient) holds
instead (tests/golden/make_golden.py::save, tests/golden/compact.py):
  key@s   every STRIDE-th element of the flattened tensor (STRIDE = 7: co-prime with every power-of-two extent, so the sample walks
          through all rows AND all columns),
  key@c   float64 [sum, sum |x|, l2 norm, max |x|, P projections on fixed +-1 vectors],
  key@m   int64 [stride, ndim, *shape].
maxerr / relerr below take such a SampledRef wherever they take a tensor: the sampled elements are compared one by one, and the
projections see every element — an error confined to unsampled elements moves each projection by +-(that error), so a single
wrong element of size e shows as e / (3 sqrt(n)) in maxerr and e / (3 |ref|) in relerr, far above the 1e-5 ... 1e-3 bounds the tests use
for O(1) tensors of 3e4 ... 3e5 elements. (Random rounding noise d projects to N(0, |d|^2): the factor 3 keeps the projection term below
the elementwise one for honest noise.)
"""
import numpy as np
import torch

STRIDE = 7
NPROJ = 8


def sign_vectors(n, name):
    """NPROJ fixed +-1 vectors of length n, a pure function of (n, name): SplitMix64 of the element index, in wrapping uint64"""
    import zlib
    seed = np.uint64(zlib.crc32(name.encode()) + 1)
    out = np.empty((NPROJ, n), np.float64)
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for p in range(NPROJ):
            x = (i + seed * np.uint64(0x9E3779B97F4A7C15) + np.uint64(p) * np.uint64(0xD1B54A32D192ED03)) * np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(30)
            x *= np.uint64(0x94D049BB133111EB)
            x ^= x >> np.uint64(31)
            out[p] = 1.0 - 2.0 * ((x >> np.uint64(63)).astype(np.float64))
    return out


def _flat64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().double().numpy()
    return np.asarray(a, dtype=np.float64).reshape(-1)


def compact_arrays(name, arr):
    """the three stored arrays of one tensor (see the module docstring); name: the key, which seeds the projection vectors"""
    a = _flat64(arr)
    chk = np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), np.abs(a).max()], sign_vectors(a.size, name) @ a])
    return {name + "@s": np.ascontiguousarray(np.asarray(arr).reshape(-1)[::STRIDE]),
            name + "@c": chk.astype(np.float64),
            name + "@m": np.array([STRIDE, np.asarray(arr).ndim, *np.asarray(arr).shape], np.int64)}


class SampledRef:
    """a reference tensor in compact form"""

    def __init__(self, name, sample, chk, meta):
        self.name = name
        self.stride = int(meta[0])
        self.shape = tuple(int(v) for v in meta[2:2 + int(meta[1])])
        self.numel = int(np.prod(self.shape)) if self.shape else 1
        self.sample = np.asarray(sample, dtype=np.float64).reshape(-1)
        self.sum, self.abssum, self.l2, self.absmax = (float(v) for v in chk[:4])
        self.proj = np.asarray(chk[4:], dtype=np.float64)
        assert self.sample.size == (self.numel + self.stride - 1) // self.stride, name

    # what the tests ask of a reference tensor besides the comparisons
    def norm(self):
        return self.l2

    def __array__(self, *a, **k):
        raise TypeError(f"golden tensor {self.name!r} is stored in compact form (tests/compare.py): compare with maxerr / relerr")

    def _diffs(self, a):
        a = _flat64(a)
        assert a.size == self.numel, (self.name, a.size, self.shape)
        d = a[::self.stride] - self.sample
        p = np.abs(sign_vectors(a.size, self.name) @ a - self.proj).max() / 3.0
        return d, p

    def maxerr(self, a):
        d, p = self._diffs(a)
        return float(max(np.abs(d).max(), p / np.sqrt(self.numel)))

    def relerr(self, a):
        d, p = self._diffs(a)
        ns = np.sqrt((self.sample * self.sample).sum())
        return float(max(np.sqrt((d * d).sum()) / max(ns, 1e-30), p / max(self.l2, 1e-30)))


def relerr(a, b):
    """|a - b| / |b| (b: the reference; a tensor, an array or a SampledRef)"""
    if isinstance(b, SampledRef):
        return b.relerr(a)
    a, b = _flat64(a), _flat64(b)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))


def maxerr(a, b):
    if isinstance(b, SampledRef):
        return b.maxerr(a)
    return float(np.abs(_flat64(a) - _flat64(b)).max())


def absmax(b):
    return b.absmax if isinstance(b, SampledRef) else float(np.abs(_flat64(b)).max())


def l2norm(b):
    return b.l2 if isinstance(b, SampledRef) else float(np.sqrt((_flat64(b) ** 2).sum()))


def allclose(a, b, atol=1e-8, rtol=1e-5):
    """torch.allclose(a, b, atol, rtol) with b the reference; for a SampledRef: elementwise on the sample, and every projection of the
    error within the same bound taken at the reference's rms (see the module docstring)"""
    if isinstance(b, SampledRef):
        d, p = b._diffs(a)
        ok = bool((np.abs(d) <= atol + rtol * np.abs(b.sample)).all())
        return ok and p / np.sqrt(b.numel) <= atol + rtol * b.l2 / np.sqrt(b.numel)
    a, b = _flat64(a), _flat64(b)
    return bool((np.abs(a - b) <= atol + rtol * np.abs(b)).all())
