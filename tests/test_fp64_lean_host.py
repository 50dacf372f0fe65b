"""CPU-side checks of the arithmetic claims in chess2rt_amd/csrc/fp64_lean.h (the device check proper is
tests/fp64_lean_check.hip, run under -m gpu): the integer re-normalisation of unit vectors, the window test and
Markstein's correction step, restated in numpy on IEEE doubles.  No GPU, no oracle."""
import struct

import numpy as np

ONE = 0x3FF0000000000000


def f(bits):
    return struct.unpack("<d", struct.pack("<Q", bits & 0xFFFFFFFFFFFFFFFF))[0]


def b(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def test_unit_len_integer_formula_equals_ieee_sqrt_and_reciprocal():
    """unit_len: for s = ONE + k (k in -64 .. 64 ulps): len = ONE + (k >> 1); with dl = k >> 1:
    inv = ONE - 2 dl (dl >= 0) or ONE + ((1 - dl) >> 1) (dl < 0) — exactly RN(sqrt s) and RN(1 / RN(sqrt s))."""
    for k in range(-64, 65):
        s = f(ONE + k)
        dl = k >> 1
        di = -2 * dl if dl >= 0 else (1 - dl) >> 1
        want_len = np.sqrt(np.float64(s))
        want_inv = np.float64(1.0) / want_len
        assert f(ONE + dl) == want_len, k
        assert f(ONE + di) == want_inv, k


def in_window(x, lo=-240, hi=240):
    hi32 = (b(x) >> 32) & 0xFFFFFFFF
    t = ((hi32 << 1) - ((lo + 1023) << 21)) & 0xFFFFFFFF
    return t < ((hi - lo) << 21)


def test_window_test_accepts_exactly_the_finite_magnitudes_in_range():
    inside = [1.0, -1.0, 2.0 ** -240, -(2.0 ** -240), np.nextafter(2.0 ** 240, 0), 1e-72, 1e72, 123.456, -9e50]
    outside = [0.0, -0.0, 5e-324, 2.2250738585072014e-308, np.nextafter(2.0 ** -240, 0), 2.0 ** 240, -(2.0 ** 240),
               1e80, 1e300, float("inf"), float("-inf"), float("nan")]
    for x in inside:
        assert in_window(x), x
    for x in outside:
        assert not in_window(x), x
    # the running unsigned maximum of t flags a sequence iff one member is outside (Oob in c2rt_trace.inc)
    rng = np.random.RandomState(5)
    span = (240 + 240) << 21
    for _ in range(2000):
        xs = list(np.ldexp(rng.uniform(1, 2, 6), rng.randint(-300, 300, 6)) * rng.choice([-1, 1], 6))
        ts = [(((b(x) >> 32) & 0xFFFFFFFF) * 2 - ((-240 + 1023) << 21)) & 0xFFFFFFFF for x in xs]
        assert (max(ts) >= span) == (not all(in_window(x) for x in xs))


def test_correction_step_through_the_rounded_reciprocal_is_the_ieee_quotient():
    """div_with(a, b, y) with y = RN(1 / b): q = RN(a y); rem = a - b q (one rounding: an fma);
    RN(q + rem y) == RN(a / b) — checked here with exact rational arithmetic for the fma."""
    from fractions import Fraction

    def fma(x, y, z):
        return np.float64(float(Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))))  # float(Fraction) rounds to nearest

    rng = np.random.RandomState(11)
    a = np.ldexp(rng.uniform(1, 2, 4000), rng.randint(-40, 40, 4000)) * rng.choice([-1, 1], 4000)
    d = np.ldexp(rng.uniform(1, 2, 4000), rng.randint(-30, 30, 4000)) * rng.choice([-1, 1], 4000)
    d[:50] = np.ldexp(np.nextafter(2.0, 0), rng.randint(-30, 30, 50))   # significands of all ones
    for x, y in zip(a, d):
        r = np.float64(1.0) / np.float64(y)
        q = np.float64(x) * r
        rem = fma(-y, q, x)
        got = fma(rem, r, q)
        assert got == np.float64(x) / np.float64(y), (x, y)
