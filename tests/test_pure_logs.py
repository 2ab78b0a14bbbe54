"""The premise of the tabulated logs (csrc/callmath.h PT_*): when one class of a strand's pair is empty,
get_Z's three quotients (reference src/genotype_model.c:23-42) lie beyond the clamp by a wide margin whatever the counts, so
the three Z are the constants 0.5 * ((l - t) * -+1 + 2 - (l + t)) and the class's log arguments depend on its quality index
alone.  Checked here on the formula itself, in float64 as the kernels and the reference evaluate it, over the whole parameter
range bsc_create accepts."""
import numpy as np


def _quotients(x1, x2, k1, k2, l, t):
    lpt, lmt = l + t, l - t
    d = (x1 + x2) * lmt
    a2 = 2.0 - lpt
    n0 = x1 * (lpt + 2.0 * k2) - x2 * (a2 + 2.0 * k1)
    n1 = x1 * (2.0 + lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1)
    n2 = x1 * (lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1)
    return n0 / d, n1 / d, n2 / d


def test_quotients_of_a_pair_with_an_empty_class_are_beyond_the_clamp():
    rng = np.random.default_rng(12)
    n = 2_000_000
    under = rng.choice([0.0, 1e-9, 0.01, 0.3, 0.999999], size=n) * rng.random(n) ** rng.integers(0, 3, size=n)
    over = rng.random(n) * (1.0 - under - 2.0 ** -20)
    over[rng.random(n) < 0.1] = 0.0
    l, t = 1.0 - under, over
    assert ((l - t) >= 2.0 ** -20).all() and (l <= 1.0).all() and (t >= 0.0).all()
    cnt = np.floor(2.0 ** (33 * rng.random(n))).astype(np.float64)  # 1 .. 2^33
    k = 0.5 * rng.random(n)  # q_prob[q].k is in (0, 0.5]
    k[rng.random(n) < 0.05] = 0.5
    kz = np.full(n, 0.5)  # the empty class: q_prob[0].k
    zero = np.zeros(n)
    for s in _quotients(zero, cnt, kz, k, l, t):  # first class of the pair empty (x1 = 0): clamped to -1
        assert (s <= -2.0 + 1e-9).all(), s.max()
    for s in _quotients(cnt, zero, k, kz, l, t):  # second class empty (x2 = 0): clamped to +1
        assert (s >= 2.0 - 1e-9).all(), s.min()
