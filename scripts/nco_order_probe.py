"""Development: bit-exact fraction / worst ulp of the GPU NCO against the oracle at growing |place| (the first/second-order
switch in plan_init and qd_shift)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import numpy as np
import quadrs_amd as Q
from oracle import oracle as O
ratio = O.shift_ratio(280000, 21_000_000)
n = 1 << 21
x = np.ones((n, 2), np.float32); x[:, 1] = 0.0          # multiplier itself: (1 + 0i) * mul
for lg in (20, 26.9, 27.5, 28.5, 28.99, 29.2, 31, 33):
    off = int((2.0 ** lg) / abs(ratio)) - n // 2
    off = max(off, 0)
    ref = O.shift_apply(x, off, ratio)
    got = Q.shift(x.copy(), off, ratio)
    exact = (ref.view(np.uint32) == got.view(np.uint32)).all(axis=1).mean()
    d = np.abs(ref.astype(np.float64) - got.astype(np.float64)).max()
    mism = int((ref.view(np.uint32) != got.view(np.uint32)).any(axis=1).sum())
    print(f"|place| ~ 2^{lg}: exact fraction {exact:.7f} ({mism} of {n} samples differ), max abs diff {d:.3e}", flush=True)
