"""The QUAL staircase (bsc_phred_table; csrc/bscall_api.c: bsc_build_phred_table): the fused chain does not evaluate
src/print_vcf.c:140-148's log for the record's QUAL but looks om = 1 - z up between the steps of the staircase
phred(om) = min(255, (int)(-10 * log(om) / LOG10)).  Host-only, no GPU.  Checked here against libm — Python's math.log is
the C library's log, the reference's — double by double around every step and on random arguments:
  * the lookup equals the direct evaluation for every double within 4 096 ulps of each of the 255 steps (log's error, below
    1 ulp of its result, moves a step by at most a couple of hundred ulps of om: outside that neighbourhood of a step the integer part
    cannot depend on the rounding, inside it every double is tried);
  * and for two million random om over the binades that occur (om is a multiple of 2^-53), plus the ends of every binade.
The device side of the lookup is exercised by every record parity test (the oracle evaluates the log)."""
import ctypes as C
import math
import struct

import numpy as np

from bs_call_amd import _lib
from oracle import loader

LOG10 = 2.30258509299404568402


def direct(om):
    p = int(-10.0 * math.log(om) / LOG10)
    return 255 if p > 255 else p


def table():
    L = _lib.load()
    thr = np.zeros((64, 4), dtype=np.float64)
    base = np.zeros(64, dtype=np.uint8)
    assert L.bsc_phred_table(thr.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p)) == 0
    return thr, base


def lookup(thr, base, om):
    e = 1023 - ((struct.unpack("<Q", struct.pack("<d", om))[0] >> 52) & 0x7FF)
    e = min(e, 63)
    return int(base[e]) + sum(1 for j in range(4) if om <= thr[e][j])


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def from_bits(u):
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def test_table_shape():
    thr, base = table()
    assert base[0] == 0 and base[53] == direct(from_bits(bits(2.0**-52) - 1))
    steps = sorted({float(t) for t in thr.reshape(-1) if t > 0}, reverse=True)
    assert len(steps) >= 160  # the steps up to 2^-53's QUAL (159) are all there ...
    for e in range(64):  # ... and a binade's four are its next four
        row = [t for t in thr[e] if t > 0]
        assert row == sorted(row, reverse=True)
        assert all(t <= 2.0 ** (1 - e) for t in row)


def test_every_double_around_every_step():
    if not loader.libm_exact():
        import pytest

        pytest.skip("this host's libm is not the glibc >= 2.28 FMA build the kernels reproduce")
    thr, base = table()
    steps = sorted({float(t) for t in thr.reshape(-1) if t > 0})
    checked = 0
    for t in steps:
        b = bits(t)
        for u in range(b - 4096, b + 4097):
            om = from_bits(u)
            if not (0.0 < om <= 1.0):
                continue
            assert lookup(thr, base, om) == direct(om), (om.hex(), t.hex())
            checked += 1
    assert checked > 1_000_000


def test_random_arguments_and_binade_ends():
    if not loader.libm_exact():
        import pytest

        pytest.skip("this host's libm is not the glibc >= 2.28 FMA build the kernels reproduce")
    thr, base = table()
    rng = np.random.default_rng(11)
    # om = 1 - z for doubles z in [0, 1): a multiple of 2^-53; sample z log-uniformly close to 1 as well as anywhere
    zs = np.concatenate([rng.random(600_000), 1.0 - np.exp(rng.uniform(math.log(2.0**-53), 0.0, 1_400_000))])
    for z in zs:
        om = 1.0 - float(z)
        if om > 0.0:
            assert lookup(thr, base, om) == direct(om), om.hex()
    for e in range(0, 54):
        for om in (2.0**-e, from_bits(bits(2.0**-e) + 1), from_bits(bits(2.0 ** (1 - e)) - 1) if e else 1.0):
            assert lookup(thr, base, om) == direct(om), om.hex()
