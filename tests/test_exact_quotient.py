"""The two-sweep kernel forms div / n (n = 1..6 Fluid neighbours, advection-reflection/main.cpp:222) in single
precision as float(double(div) * (1.0 / n)).  That must be the IEEE quotient for every float: checked here on the
host on a sample of bit patterns that covers every exponent (subnormals, zeros, infinities and NaNs included);
scripts/check_exact_quotient.py runs all 2^32 patterns for one divisor in about a minute."""
import numpy as np
import pytest


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_product_with_the_double_reciprocal_is_the_float_quotient(n):
    rng = np.random.default_rng(n)
    bits = rng.integers(0, 2 ** 32, size=1 << 21, dtype=np.uint64).astype(np.uint32)
    # every exponent with a few mantissas around the ends, both signs
    exps = np.arange(256, dtype=np.uint32)[:, None] << 23
    mants = np.array([0, 1, 2, 3, 0x2aaaaa, 0x2aaaab, 0x400000, 0x555555, 0x7ffffd, 0x7ffffe, 0x7fffff], dtype=np.uint32)[None, :]
    edge = (exps | mants).ravel()
    bits = np.concatenate([bits, edge, edge | np.uint32(0x80000000)])
    x = bits.view(np.float32)
    with np.errstate(all="ignore"):
        want = x / np.float32(n)
        got = (x.astype(np.float64) * (np.float64(1.0) / np.float64(n))).astype(np.float32)
    same = (want.view(np.uint32) == got.view(np.uint32)) | (np.isnan(want) & np.isnan(got))
    assert same.all(), (n, x[~same][:4], want[~same][:4], got[~same][:4])
