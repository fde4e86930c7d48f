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


def test_fmod_is_exact(orc):
    x = np.concatenate([rng.uniform(-100, 100, 200000), rng.uniform(-1e9, 1e9, 50000), [0.0, 2 * np.pi, -2 * np.pi, 7.0]])
    assert np.array_equal(orc.detmath(4, x), orc.libm(4, x))
