"""The kernels turn an 8-bit texel channel into a float with a product and one residual step
instead of a division (rt_shading.h, unorm8).  For every byte that must be the correctly rounded
byte / 255.0f the reference's UNORM sampler and the oracle produce."""
import numpy as np


def fma32(a, b, c):
    # exact in float64: |a b| < 2^9 with 48 significant bits, c within a few binades of it
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def test_unorm8_equals_the_division_for_every_byte():
    r = np.float32(1.0) / np.float32(255.0)
    for byte in range(256):
        x = np.float32(byte)
        q = np.float32(x * r)
        got = fma32(fma32(-q, np.float32(255.0), x), r, q)
        assert got == np.float32(x / np.float32(255.0)), byte
