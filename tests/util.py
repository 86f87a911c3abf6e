import numpy as np

import os

# /root/reference/README.md:167, copied verbatim as an expected-output fixture
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_ook_line167.txt")) as _f:
    README_OOK = _f.read().strip()


def ook_pipeline(text):
    """The README's sed/tr pipeline (README.md:122-167) over spark_fft's stdout bytes."""
    import re
    bits = []
    for line in text.decode("utf-8").split("\n")[:-1]:
        line = re.sub(r"^.    .$", ".", line)
        line = re.sub(r"....*", "X", line)
        bits.append(line)
    s = "".join(bits).replace(".", "o")     # the README writes '.' in one step and 'o' in the next
    return re.sub(r"o{5,10}", "B", re.sub(r"X{6,10}", "A", s))


def ulp_diff(a, b):
    """distance in units of f32 representable steps (sign-magnitude ordered)"""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def ulp_of(x):
    """spacing of f32 at |x|"""
    x = np.abs(np.asarray(x, dtype=np.float32))
    return np.spacing(np.maximum(x, np.float32(1e-45)))


def complex_ulp_err(ref, got):
    """|got - ref| per component in units of ulp(max(|re|,|im|)) of the reference sample —
    the cf32 tolerance unit used throughout (north_star: 'within 1 ulp on cf32')."""
    ref = np.asarray(ref, dtype=np.float32).reshape(-1, 2)
    got = np.asarray(got, dtype=np.float32).reshape(-1, 2)
    scale = ulp_of(np.max(np.abs(ref), axis=1)).astype(np.float64)
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max(axis=1)
    return d / scale


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
