"""The HIP unpack computes `sample / 127.0`, `/ 255.0`, `/ 65535.0` (src/samples.rs:93-127 via the oracle's restatement) without
a division: 1/d = hi + lo (hi = RN(1/d), lo = RN(1/d - hi)), result = fma(f, hi, RN(f * lo)) (quadrs_amd/csrc/qd_device.h:
div_small, with the constants written there as hex floats).  This restates the two operations in exact rational arithmetic and checks
them against the correctly rounded quotient for EVERY input value of the three integer sample formats — the kernel's bit-exactness for
cs8 / cu8 / cs16 rests on it — and the header's literal constants against hi and lo."""
import os
import re
import math
from fractions import Fraction


def rn32(x):
    """Fraction -> nearest-even binary32 value (normal range), as a Fraction."""
    if x == 0:
        return Fraction(0)
    s = 1 if x > 0 else -1
    a = abs(Fraction(x))
    e = math.floor(math.log2(a))
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    q = a / ulp
    n = math.floor(q)
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return s * n * ulp


def split(d):
    hi = rn32(Fraction(1, d))
    return hi, rn32(Fraction(1, d) - hi)


def div_small(f, d):
    hi, lo = split(d)
    t = rn32(f * lo)             # the multiply: one rounding
    return rn32(f * hi + t)      # fma(f, hi, t): one rounding


def test_header_constants_are_the_split_reciprocals():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quadrs_amd", "csrc", "qd_device.h")).read()
    for d in (127, 255, 65535):
        m = re.search(r"kInv%dHi = (-?0x[0-9a-fp.+-]+)f, kInv%dLo = (-?0x[0-9a-fp.+-]+)f" % (d, d), hdr)
        assert m, d
        hi, lo = split(d)
        assert Fraction(float.fromhex(m.group(1))) == hi and Fraction(float.fromhex(m.group(2))) == lo, (d, m.groups())


def test_reciprocal_plus_residual_equals_ieee_division_for_every_sample_value():
    for values, d in ((range(-128, 128), 127), (range(0, 256), 255), (range(-32768, 32768), 65535)):
        bad = [i for i in values if div_small(Fraction(i), d) != rn32(Fraction(i, d))]
        assert not bad, (d, bad[:8])
