"""The compact form of a golden tensor (tests/compare.py) must see what the full tensor saw: an error in ANY element, sampled or not."""
import numpy as np
import torch

from compare import STRIDE, SampledRef, allclose, compact_arrays, maxerr, relerr, sign_vectors


def _ref(n=32768, seed=0):
    a = np.random.default_rng(seed).standard_normal(n).astype(np.float32).reshape(128, -1)
    c = compact_arrays("k", a)
    return a, SampledRef("k", c["k@s"], c["k@c"], c["k@m"])


def test_sign_vectors_are_a_pure_function_and_balanced():
    v = sign_vectors(4099, "some.key")
    assert np.array_equal(v, sign_vectors(4099, "some.key")) and set(np.unique(v)) == {-1.0, 1.0}
    assert not np.array_equal(v, sign_vectors(4099, "other.key")) and (np.abs(v.sum(1)) < 5 * np.sqrt(4099)).all()
    assert np.abs(v @ v.T / 4099 - np.eye(len(v))).max() < 0.08           # the eight vectors are not copies of each other


def test_equal_tensors_and_rounding_noise_pass_like_the_full_comparison():
    a, r = _ref()
    assert maxerr(torch.from_numpy(a), r) == 0.0 and relerr(a, r) == 0.0 and allclose(a, r, atol=0.0, rtol=0.0)
    noisy = a + 1e-6 * np.random.default_rng(1).standard_normal(a.shape).astype(np.float32)
    full_max, full_rel = maxerr(noisy, a), relerr(noisy, a)
    assert 0.5 * full_max < maxerr(noisy, r) <= 1.05 * full_max          # the sampled maximum, never above the true one by the projection term
    assert 0.8 * full_rel < relerr(noisy, r) < 1.2 * full_rel
    assert allclose(noisy, r, atol=1e-5) and not allclose(noisy, r, atol=1e-7, rtol=0.0)


def test_one_wrong_element_is_seen_wherever_it_sits():
    a, r = _ref()
    for pos in (0, 1, STRIDE - 1, STRIDE, 12345, a.size - 1):            # sampled and unsampled positions
        b = a.copy().reshape(-1)
        b[pos] += 0.25
        assert maxerr(b, r) > 2e-4 and relerr(b, r) > 2e-4 and not allclose(b, r, atol=1e-4, rtol=1e-4), pos
    b = a.copy()
    b[17, 3:40] *= 1.01                                                   # part of one row off by 1 %
    assert maxerr(b, r) > 1e-3


def test_wrong_size_is_refused():
    a, r = _ref()
    try:
        maxerr(a.reshape(-1)[:-1], r)
    except AssertionError:
        return
    raise AssertionError("a tensor of another size compared without complaint")
