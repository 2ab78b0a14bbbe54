"""bsmath.h (the log/exp/lgamma the kernels use) against the host libm, on the CPU.

On glibc >= 2.28 / x86-64 with FMA the replica must equal libm bit for bit; elsewhere (libm_exact False)
it must still be within 1 ulp.  Also: the two oracle flavours agree on whole gt_meth records."""
import numpy as np
import pytest

import bs_call_amd as B

# Every test here runs twice: unmarked in this container, and marked `gpu` so that the driver's GPU box — the host
# whose libm the GPU parity tests compare against — re-proves bsm == libm for itself.
BOTH_HOSTS = pytest.mark.parametrize("where", ["here", pytest.param("gpu_box", marks=pytest.mark.gpu)])


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


@BOTH_HOSTS
def test_log_exp_vs_libm(oracle, libm_exact, where):
    rng = np.random.default_rng(2024)
    n = 2_000_000
    for x in _log_inputs(rng, n):
        u = _ulps(oracle.log_array(x, oracle.LIBM), oracle.log_array(x, oracle.BSM))
        assert u.max() <= (0 if libm_exact else 1), (u.max(), x[np.argmax(u)])
    for x in _exp_inputs(rng, n):
        u = _ulps(oracle.exp_array(x, oracle.LIBM), oracle.exp_array(x, oracle.BSM))
        assert u.max() <= (0 if libm_exact else 1), (u.max(), x[np.argmax(u)])


@BOTH_HOSTS
def test_special_values(oracle, where):
    L = oracle.lib()
    assert L.orc_log(0.0, 1) == -np.inf and L.orc_log(-0.0, 1) == -np.inf
    assert np.isnan(L.orc_log(-1.0, 1)) and np.isnan(L.orc_log(np.nan, 1)) and L.orc_log(np.inf, 1) == np.inf
    assert L.orc_log(1.0, 1) == 0.0 and not np.signbit(L.orc_log(1.0, 1))
    assert L.orc_exp(-np.inf, 1) == 0.0 and L.orc_exp(np.inf, 1) == np.inf and np.isnan(L.orc_exp(np.nan, 1))
    assert L.orc_exp(0.0, 1) == 1.0 and L.orc_exp(-1e4, 1) == 0.0 and L.orc_exp(1e4, 1) == np.inf


@BOTH_HOSTS
def test_lfact_vs_libm(oracle, tables, libm_exact, where):
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


@BOTH_HOSTS
def test_oracle_flavours_agree_on_records(oracle, tables, libm_exact, where):
    """Whole gt_meth records: libm flavour (the reference's arithmetic) == bsm flavour (the kernels' arithmetic)."""
    if not libm_exact:
        pytest.skip("host libm is not glibc's FMA variant")
    for cov, n in ((30, 300_000), (300, 20_000)):
        pile, ref = B.synth_pileup_host(99 + cov, 0, n, cov)
        a, sa = oracle.call_sites(pile, ref, tables, oracle.LIBM, -8)
        b, sb = oracle.call_sites(pile, ref, tables, oracle.BSM, -8)
        assert a.tobytes() == b.tobytes() and (sa == sb).all()


@BOTH_HOSTS
def test_oracle_threading_modes_agree(oracle, tables, where):
    pile, ref = B.synth_pileup_host(5, 0, 50_000, 30)
    a, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, 1)
    b, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, 5)  # the reference's interleaved striding
    c, _ = oracle.call_sites(pile, ref, tables, oracle.LIBM, -3)
    assert a.tobytes() == b.tobytes() == c.tobytes()


@pytest.mark.gpu
def test_gpu_box_libm_is_exact(oracle, libm_exact):
    """On the GPU box the LIBM flavour (the reference's arithmetic on the host's libm, no product code in it) must be
    the asserted one: an x86-64 glibc >= 2.28 host with FMA selects glibc's *_fma log/exp, which bsmath.h replicates
    bit for bit.  If this fails the parity tests silently fell back to integers-exact + 1e-11 against LIBM."""
    import platform

    host = oracle.host_description()
    print("libm_exact=%s host=%s" % (libm_exact, host))
    name, ver = platform.libc_ver()
    glibc_ok = name == "glibc" and tuple(int(v) for v in ver.split(".")[:2]) >= (2, 28)
    if platform.machine() == "x86_64" and glibc_ok and host["cpu_fma"]:
        assert libm_exact, "x86-64 glibc %s with FMA, yet libm log/exp differ from bsmath.h" % ver
    else:
        pytest.skip("not an x86-64 glibc >= 2.28 host with FMA: %s" % host)
