import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import quadrs_amd as Q
from oracle import oracle as O
ratio = Q.shift_ratio(280000, 21000000)
n = 1 << 20
x = np.zeros((n, 2), np.float32); x[:, 0] = 1.0
for off in (0, 12345, 2**31 - 777, 2**33 + 5):
    a = O.shift_apply(x, off, ratio); b = Q.shift(x, off, ratio)
    bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
    print("off", off, "mismatches", bad.size, "of", n)
    for i in bad[:12]:
        nn = off + int(i)
        place = float(nn) * ratio
        print("   n", nn, "row", nn // 512, "j", nn % 512, "oracle", a[i], "gpu", b[i], "place", repr(place),
              "cos64", repr(np.cos(place)), "sin64", repr(np.sin(place)))
