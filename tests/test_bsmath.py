"""bsmath.h (the log/exp/lgamma the kernels use) against the host libm, on the CPU.

On glibc >= 2.28 / x86-64 with FMA the replica must equal libm bit for bit; elsewhere (libm_exact False)
it must still be within 1 ulp.  Also: the two oracle flavours agree on whole gt_meth records."""
import numpy as np
import pytest

import bs_call_amd as B


def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def _log_inputs(rng, n):
    return [
        rng.uniform(1e-5, 2.2, n),  # arguments of the Z terms: k .. 2 + k
        rng.uniform(1.0, 10.0, n),  # log(sum)
        rng.uniform(0.93, 1.07, n),  # the near-1 branch and its edges
        np.exp(rng.uniform(-744, 709, n)),  # whole range
        np.abs(rng.standard_normal(n)) * 1e-310,  # subnormals
        np.array([1.0, 0.9375, 1.064697265625, 2.0, 0.5, 1e-20, 257.0, 1e300, 5e-324, 2.2250738585072014e-308]),
    ]


def _exp_inputs(rng, n):
    return [
        rng.uniform(-745.2, 0, n),
        rng.uniform(-40, 0, n),
        rng.uniform(0, 709.8, n),
        rng.uniform(-1e-3, 1e-3, n),
        rng.uniform(-750, -700, n),  # subnormal results, underflow
        rng.uniform(700, 711, n),  # overflow edge
        rng.standard_normal(n) * 1e-17,
        np.array([0.0, -0.0, 1.0, -1.0, -708.4, -745.13, -745.14, 709.78, 709.79, -1e308, 1e308, 512.0, -512.0, -1024.0]),
    ]


def test_log_exp_vs_libm(oracle, libm_exact):
    rng = np.random.default_rng(2024)
    n = 2_000_000
    for x in _log_inputs(rng, n):
        u = _ulps(oracle.log_array(x, oracle.LIBM), oracle.log_array(x, oracle.BSM))
        assert u.max() <= (0 if libm_exact else 1), (u.max(), x[np.argmax(u)])
    for x in _exp_inputs(rng, n):
        u = _ulps(oracle.exp_array(x, oracle.LIBM), oracle.exp_array(x, oracle.BSM))
        assert u.max() <= (0 if libm_exact else 1), (u.max(), x[np.argmax(u)])


def test_special_values(oracle):
    L = oracle.lib()
    assert L.orc_log(0.0, 1) == -np.inf and L.orc_log(-0.0, 1) == -np.inf
    assert np.isnan(L.orc_log(-1.0, 1)) and np.isnan(L.orc_log(np.nan, 1)) and L.orc_log(np.inf, 1) == np.inf
    assert L.orc_log(1.0, 1) == 0.0 and not np.signbit(L.orc_log(1.0, 1))
    assert L.orc_exp(-np.inf, 1) == 0.0 and L.orc_exp(np.inf, 1) == np.inf and np.isnan(L.orc_exp(np.nan, 1))
    assert L.orc_exp(0.0, 1) == 1.0 and L.orc_exp(-1e4, 1) == 0.0 and L.orc_exp(1e4, 1) == np.inf


def test_lfact_vs_libm(oracle, tables, libm_exact):
    """lfact2 (include/bs_call.h:335): table below 256, lgamma(x + 1) from 256 on."""
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = list(range(0, 20_000)) + [int(v) for v in rng.integers(20_000, 2**31 - 2, 20_000)]
    for x in xs:
        a, b = L.orc_lfact(x, tables.ptr, 0), L.orc_lfact(x, tables.ptr, 1)
        if libm_exact or x < 256:
            assert a == b, x
        else:
            assert abs(a - b) <= 4e-16 * abs(a), x


def test_oracle_flavours_agree_on_records(oracle, tables, libm_exact):
    """Whole gt_meth records: libm flavour (the reference's arithmetic) == bsm flavour (the kernels' arithmetic)."""
    if not libm_exact:
        pytest.skip("host libm is not glibc's FMA variant")
    for cov, n in ((30, 300_000), (300, 20_000)):
        pile, ref = B.synth_pileup_host(99 + cov, 0, n, cov)
        a, sa = oracle.call_sites(pile, ref, tables, oracle.LIBM, -8)
        b, sb = oracle.call_sites(pile, ref, tables, oracle.BSM, -8)
        assert a.tobytes() == b.tobytes() and (sa == sb).all()


def test_oracle_threading_modes_agree(oracle, tables):
    pile, ref = B.synth_pileup_host(5, 0, 50_000, 30)
    a, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, 1)
    b, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, 5)  # the reference's interleaved striding
    c, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, -3)
    assert a.tobytes() == b.tobytes() == c.tobytes()
