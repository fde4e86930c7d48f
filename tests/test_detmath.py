"""detmath.h (the functions the HIP kernels call) against 80-bit long double and glibc."""
import numpy as np
import pytest

rng = np.random.default_rng(1234)


def ulp_err(got, x, f):
    ref = f(x.astype(np.longdouble))
    u = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    return float(np.max(np.abs(got.astype(np.longdouble) - ref) / u))


@pytest.mark.parametrize("fn,f", [(0, np.sin), (1, np.cos)])
def test_sincos_within_one_ulp(orc, fn, f):
    x = np.concatenate([rng.uniform(-10, 10, 400000), rng.uniform(-2000, 2000, 100000),
                        np.arange(-16, 17) * (np.pi / 4), [1e-300, 0.0, -0.0, 1e5, 1048575.5]])
    assert ulp_err(orc.detmath(fn, x), x, f) < 1.0
    # same order of accuracy as glibc; last-bit disagreements are expected and rare
    assert np.mean(orc.detmath(fn, x) != orc.libm(fn, x)) < 0.06


@pytest.mark.parametrize("fn,f", [(0, np.sin), (1, np.cos)])
def test_sincos_table_seams_zeros_and_hand_over(orc, fn, f):
    """The table-driven sin / cos of round 3 (|x| < 16384): half a unit in the last place plus the final rounding's slack, also
    where it is hardest: either side of the seams between table entries, next to the multiples of pi/128 (among them the zeros
    of sin and cos, where the reduction is redone with more of pi), the band next to a zero entry (where the table's tail
    times r counts), tiny arguments; the polynomial form beyond 16384 stays within one unit."""
    h = np.pi / 128
    k = np.arange(0, 700, dtype=np.float64)
    x = np.concatenate([rng.uniform(-16384, 16384, 200000), rng.uniform(-8, 8, 200000),
                        rng.uniform(0.5 * h, 1.5 * h, 50000), np.pi / 2 + rng.uniform(-2 * h, 2 * h, 50000), np.pi + rng.uniform(-2 * h, 2 * h, 50000)]
                       + [(k + 0.5) * h * (1 + e) for e in (0.0, 1e-16, -1e-16, 1e-9, -1e-9)]
                       + [k * h * (1 + e) for e in (0.0, 1.2e-16, -1.2e-16, 3e-13, -3e-13, 1e-8)]
                       + [10.0 ** rng.uniform(-300, -2, 20000), -(10.0 ** rng.uniform(-20, -2, 20000)), [0.0, 16383.999999999998]])
    assert ulp_err(orc.detmath(fn, x), x, f) < 0.502
    z = orc.detmath(fn, [0.0, -0.0])
    assert (z[0] == 0.0 and z[1] == 0.0 and np.signbit(z[1]) and not np.signbit(z[0])) if fn == 0 else (z[0] == 1.0 and z[1] == 1.0)
    big = np.concatenate([rng.uniform(16384, 1048576, 50000), -rng.uniform(16384, 1048576, 50000), [16384.0, -16384.0]])
    assert ulp_err(orc.detmath(fn, big), big, f) < 1.0
    assert np.all(np.isnan(orc.detmath(fn, [np.inf, -np.inf, np.nan])))
    # odd / even in the argument, bit for bit
    y = rng.uniform(0, 40, 10000)
    a, b = orc.detmath(fn, y), orc.detmath(fn, -y)
    assert np.array_equal(a, -b if fn == 0 else b)


def test_cos_half_pi_is_accurate(orc):
    assert orc.detmath(1, [np.pi / 2])[0] == pytest.approx(6.123233995736766e-17, rel=1e-15)


def test_exp_log(orc):
    x = np.concatenate([rng.uniform(-700, 700, 300000), rng.uniform(-8, 0, 200000)])
    assert ulp_err(orc.detmath(2, x), x, np.exp) < 1.0
    y = np.concatenate([rng.uniform(0.1, 1.0, 300000), np.exp(rng.uniform(-700, 700, 100000))])
    assert ulp_err(orc.detmath(3, y), y, np.log) < 1.0
    # under/overflow edges used by exp(sum of logs) (slam/mcl.cpp:75)
    assert orc.detmath(2, [-746.0, -1000.0])[0] == 0.0 and orc.detmath(2, [-1000.0])[0] == 0.0
    assert np.isinf(orc.detmath(2, [710.0])[0])
    sub = orc.detmath(2, [-740.0])[0]
    assert sub == pytest.approx(np.exp(-740.0), rel=1e-3) and sub > 0
    assert orc.detmath(3, [1.0])[0] == 0.0


def test_exp_log_tables_edges_and_special_values(orc):
    """The table-driven exp / log of round 3: the seams of their tables (arguments one place either side of every sub-interval of
    log's z range and of every 2^(j/128) step of exp), the hand-over to fdlibm's log at 0.9375 and 1.0625, subnormal arguments
    and results, and the special values."""
    off = np.uint64(0x3FE6000000000000)
    edges = (off + (np.arange(129, dtype=np.uint64) << np.uint64(45)))
    z = np.concatenate([(edges + np.uint64(d)).view(np.float64) for d in (0, 1, 2)] + [(edges - np.uint64(d)).view(np.float64) for d in (1, 2)])
    y = np.concatenate([z * s for s in (1.0, 0.125, 2.0 ** -300, 2.0 ** 500, 2.0 ** -1022)] + [[0.9375, np.nextafter(0.9375, 0), 1.0625, np.nextafter(1.0625, 2),
                        np.nextafter(1.0, 0), np.nextafter(1.0, 2), 5e-324, 1e-310, 2.2250738585072014e-308, 1.7976931348623157e308]])
    assert ulp_err(orc.detmath(3, y), y, np.log) < 0.6
    near1 = rng.uniform(0.93, 1.07, 200000)
    assert ulp_err(orc.detmath(3, near1), near1, np.log) < 0.6
    k = np.arange(-137000, 131000, 37, dtype=np.float64)
    x = np.concatenate([k * (np.log(2.0) / 128) * (1 + e) for e in (0.0, 1e-16, -1e-16, 3e-3 / np.maximum(np.abs(k), 1))])
    x = x[(x > -708.0) & (x < 709.7)]
    assert ulp_err(orc.detmath(2, x), x, np.exp) < 0.52
    sub = rng.uniform(-745.0, -708.4, 50000)   # subnormal results: one more rounding
    got = orc.detmath(2, sub)
    assert np.all(np.abs(got.astype(np.longdouble) - np.exp(sub.astype(np.longdouble))) <= 1.0 * 5e-324)
    inf, nan = np.inf, np.nan
    sv = orc.detmath(3, [0.0, -0.0, -1.0, inf, nan, 1.0])
    assert sv[0] == -inf and sv[1] == -inf and np.isnan(sv[2]) and sv[3] == inf and np.isnan(sv[4]) and sv[5] == 0.0
    ev = orc.detmath(2, [0.0, -0.0, inf, -inf, nan, 1e-300, -1e-300, 709.782712893384, 709.7827128933841])
    assert ev[0] == 1.0 and ev[1] == 1.0 and ev[2] == inf and ev[3] == 0.0 and np.isnan(ev[4]) and ev[5] == 1.0 and ev[6] == 1.0
    assert np.isfinite(ev[7]) and ev[8] == inf
    # monotone where the filter uses them: the likelihood's arguments
    q = np.linspace(-8.0, 0.0, 400001)
    assert np.all(np.diff(orc.detmath(2, q)) >= 0)
    p = np.linspace(0.1, 0.3, 400001)
    assert np.all(np.diff(orc.detmath(3, p)) >= 0)


def test_fmod_is_exact(orc):
    x = np.concatenate([rng.uniform(-100, 100, 200000), rng.uniform(-1e9, 1e9, 50000), [0.0, 2 * np.pi, -2 * np.pi, 7.0]])
    assert np.array_equal(orc.detmath(4, x), orc.libm(4, x))
