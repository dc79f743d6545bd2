"""Accuracy of the hand-written device math (csrc/fastmath.hpp) through mcg_debug_eval, against
mpmath (50 digits).  Bar: <= 2 ulp for exp / sqrt, <= 3.5e-16 absolute for sin/cos, <= 4 ulp for -2 ln u
(table + cancellation next to u = 1), and the fast normal pair within 1e-14 absolute of the RNG contract evaluated in
high precision."""
import struct

import mpmath as mp
import numpy as np
import pytest

import montecarlooptionspricer_amd as mc
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu
mp.mp.dps = 50


@pytest.fixture(scope="module")
def eng():
    e = mc.PathEngine(0)
    yield e
    e.close()


def ulp_err(got, exact_mp):
    exact = float(exact_mp)
    if exact == 0.0:
        return abs(got)
    ulp = np.spacing(abs(exact))
    return float(abs(mp.mpf(got) - exact_mp) / mp.mpf(ulp))


def test_scaled_exp(eng):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-0.7, 0.7, 3000), rs.uniform(-40, 40, 1000), rs.normal(0, 0.02, 2000),
                        [0.0, 1e-300, -1e-300, 0.34657359, -0.34657359, 709.0, -740.0, 1000.0, -1000.0]])
    y = eng.debug_eval(0, x)[:, 0]
    worst = 0.0
    for xi, yi in zip(x[:-2], y[:-2]):
        worst = max(worst, ulp_err(yi, mp.e ** mp.mpf(xi)))
    assert worst <= 2.0, worst
    assert y[-2] == np.inf and y[-1] == 0.0


def test_scaled_exp_small_variants(eng):
    """The branch-free step exponentials on the ranges the host proves for them: |a| <= 0.125 (degree 9) and
    |a| <= 0.1 (degree 8)."""
    rs = np.random.RandomState(7)
    for fn, bound in ((6, 0.125), (7, 0.1)):
        x = np.concatenate([rs.uniform(-bound, bound, 3000), rs.normal(0, 0.0126, 3000).clip(-bound, bound),
                            [0.0, bound, -bound, 1e-300, 5e-324]])
        y = eng.debug_eval(fn, x)[:, 0]
        worst = max(ulp_err(yi, mp.e ** mp.mpf(xi)) for xi, yi in zip(x, y))
        assert worst <= 2.0, (fn, worst)


def test_exp2_pair(eng):
    """2^(t/256) for arguments already in units of (ln 2)/256 (the rBergomi variance factor: 256-entry table of 2^(j/256)
    times a degree-4 polynomial): both chains of the pair, <= 1.5 ulp."""
    rs = np.random.RandomState(11)
    t = np.concatenate([rs.uniform(-128, 128, 3000), rs.uniform(-8000, 8000, 2000), rs.normal(0, 400, 2000),
                        np.arange(-280.0, 280.0, 0.5), [0.0, 0.5, -0.5, 127.5, 128.5, 255.5, 256.5, -255.5, 1e-300, 256000.0,
                                                        -272000.0, 264000.0, -308000.0]])
    y = eng.debug_eval(8, t)
    worst = 0.0
    for ti, (ya, yb) in zip(t[:-2], y[:-2, :2]):
        worst = max(worst, ulp_err(ya, mp.mpf(2) ** (mp.mpf(ti) / 256)), ulp_err(yb, mp.mpf(2) ** (mp.mpf(float(ti + 96.0)) / 256)))
    assert worst <= 1.5, worst
    assert y[-2, 0] == np.inf and y[-1, 0] == 0.0


def test_neg2log(eng):
    rs = np.random.RandomState(1)
    u = np.concatenate([rs.uniform(0, 1, 4000), 1 - rs.uniform(0, 1, 1500) ** 8, rs.uniform(0, 1, 1500) ** 12,
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.6875, 0.99609375, 0.25, 1.0, 2.0 ** -52 * 1.5]])
    u = u[(u > 0) & (u <= 1)]
    y = eng.debug_eval(1, u)[:, 0]
    worst = 0.0
    for ui, yi in zip(u, y):
        worst = max(worst, ulp_err(yi, -2 * mp.log(mp.mpf(ui))))
    assert worst <= 4.0, worst


def test_sqrt_pos(eng):
    rs = np.random.RandomState(2)
    x = np.concatenate([rs.uniform(0, 80, 4000), 10.0 ** rs.uniform(-16, 2, 2000), [2.2e-16, 73.5, 1.0, 4.0]])
    y = eng.debug_eval(2, x)[:, 0]
    worst = max(ulp_err(yi, mp.sqrt(mp.mpf(xi))) for xi, yi in zip(x, y))
    assert worst <= 1.0, worst


def test_sincos_table(eng):
    """cos/sin(2 pi f) from a raw Philox word (512-entry table + small-angle series): absolute error
    <= 3.5e-16 (what Box-Muller needs: z = R cos, R <= 7.6), and the point stays on the unit circle."""
    rs = np.random.RandomState(3)
    words = rs.randint(0, 2 ** 32, size=6000, dtype=np.uint64)
    edge = np.array([0, 2 ** 32 - 1, 1 << 29, (1 << 29) - 1, 3 << 29, (5 << 29) + 255, 7 << 29, 1 << 8, 255,
                     1 << 30, (1 << 30) - 1, 1 << 31, (1 << 31) - 1, 3 << 30, (3 << 30) - 1, 1 << 23, (1 << 23) - 1],
                    dtype=np.uint64)
    words = np.concatenate([words, edge])
    y = eng.debug_eval(3, words.astype(np.float64))
    worst = 0.0
    for w, (c, s, _, _) in zip(words, y):
        ang = 2 * mp.pi * (mp.mpf(int(w) >> 8) + mp.mpf(1) / 2) / mp.mpf(2) ** 24
        worst = max(worst, float(abs(mp.mpf(c) - mp.cos(ang))), float(abs(mp.mpf(s) - mp.sin(ang))))
    assert worst <= 3.5e-16, worst
    assert np.max(np.abs(y[:, 0] ** 2 + y[:, 1] ** 2 - 1.0)) < 1e-15


def test_normal_quad_fast_matches_contract(eng):
    orc = Oracle()
    ids = np.arange(0, 5000, dtype=np.float64)
    fast = eng.debug_eval(4, ids)
    slow = eng.debug_eval(5, ids)
    want = np.array([orc.normal_quad(1, int(i), 0, 0) for i in ids])
    assert np.max(np.abs(fast - want)) < 1e-14
    eager = eng.debug_eval(9, ids)          # the four table entries requested together: the same arithmetic, the same bits
    assert np.array_equal(eager, fast)
    assert np.max(np.abs(slow - want)) < 1e-14
    # and in high precision for a few
    for i in range(0, 5000, 500):
        w = orc.philox([i, 0, 0, 0], [1, 0])
        for h in range(2):
            wa, wb = w[2 * h], w[2 * h + 1]
            u = (mp.mpf(((wb & 0xFF) << 32) | wa) + mp.mpf(1) / 2) / mp.mpf(2) ** 40
            f = (mp.mpf(wb >> 8) + mp.mpf(1) / 2) / mp.mpf(2) ** 24
            r = mp.sqrt(-2 * mp.log(u))
            assert abs(mp.mpf(fast[i, 2 * h]) - r * mp.cos(2 * mp.pi * f)) < mp.mpf("4e-15")
            assert abs(mp.mpf(fast[i, 2 * h + 1]) - r * mp.sin(2 * mp.pi * f)) < mp.mpf("4e-15")
