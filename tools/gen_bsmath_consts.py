#!/usr/bin/env python3
"""Generate the constants of bs_call_amd/csrc/bsmath.h with 80-digit arithmetic.

Run:  python tools/gen_bsmath_consts.py   (prints C hex-float literals)
Nothing here reads the reference; the values are textbook (ln 2, 1/k!, 2/(2k+1), Stirling).
"""
from decimal import Decimal, getcontext
from fractions import Fraction
import struct

getcontext().prec = 80


def to_double(d):
    """Round a Decimal/Fraction to the nearest double (ties-to-even) exactly."""
    fr = Fraction(d) if not isinstance(d, Fraction) else d
    return float(fr)  # Fraction.__float__ is correctly rounded (int/int true division)


def hexf(x):
    return float(x).hex()


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


LN2 = Decimal(2).ln()
# ln2 split: hi keeps the top 32 bits of the mantissa word (low 32 bits zero) so k*hi is exact for |k| < 2^20
ln2_d = to_double(LN2)
hi_bits = bits(ln2_d) & 0xFFFFFFFF00000000
ln2_hi = struct.unpack("<d", struct.pack("<Q", hi_bits))[0]
ln2_lo = to_double(LN2 - Decimal(ln2_hi))
print("LN2_HI", hexf(ln2_hi), "LN2_LO", hexf(ln2_lo))
print("INV_LN2", hexf(to_double(Decimal(1) / LN2)))
print("LOG atanh-series coefficients 2/(2k+1), k=1..10")
for k in range(1, 11):
    print("  L%d" % k, hexf(to_double(Fraction(2, 2 * k + 1))))
print("EXP Taylor coefficients 1/k!, k=2..13")
f = 1
for k in range(1, 14):
    f *= k
    if k >= 2:
        print("  E%d" % k, hexf(to_double(Fraction(1, f))))
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")
print("HALF_LN_2PI", hexf(to_double((2 * PI).ln() / 2)))
print("S1=1/12", hexf(to_double(Fraction(1, 12))), "S3=1/360", hexf(to_double(Fraction(1, 360))),
      "S5=1/1260", hexf(to_double(Fraction(1, 1260))))
print("LOG10(ln 10, as the reference macro 2.30258509299404568402)", hexf(2.30258509299404568402))
