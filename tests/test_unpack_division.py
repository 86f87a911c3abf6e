"""The HIP unpack computes `sample / 127.0`, `/ 255.0`, `/ 65535.0` (src/samples.rs:93-127 via the oracle's restatement) without
a division: q = f * RN(1/d), e = fma(-q, d, f), result = fma(e, RN(1/d), q) (quadrs_amd/csrc/qd_device.h: div_small).  This
restates that sequence in exact rational arithmetic and checks it against the correctly rounded quotient for EVERY input value of
the three integer sample formats — the kernel's bit-exactness for cs8 / cu8 / cs16 rests on it."""
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


def div_small(f, d):
    rd = rn32(Fraction(1, d))
    q = rn32(f * rd)
    e = rn32(f - q * d)          # fma(-q, d, f): one rounding
    return rn32(q + e * rd)      # fma(e, rd, q): one rounding


def test_reciprocal_plus_residual_equals_ieee_division_for_every_sample_value():
    for values, d in ((range(-128, 128), 127), (range(0, 256), 255), (range(-32768, 32768), 65535)):
        bad = [i for i in values if div_small(Fraction(i), d) != rn32(Fraction(i, d))]
        assert not bad, (d, bad[:8])
