"""Full-size parity census: EVERY window of a BASELINE workload at its full size, HIP chain kernel against the CPU oracle (the
cheapest exact form of the oracle, all host cores).  The GPU suite runs it for cfg2, cfg3' and cfg4
(tests/test_gpu_parity.py::test_full_size_census); cfg3 takes 1.5-3 minutes of host time and is run from here.
Test infrastructure (uses oracle/): never imported by the product.   usage: python scripts/full_census.py cfg2 cfg3p cfg3 cfg4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import quadrs_amd as Q
from oracle import oracle as O
from util import full_size_census

for name in sys.argv[1:]:
    total, kind, nw, nb, worst, first, t_gpu, t_cpu, cores = full_size_census(Q, O, bench, name)
    cfg = bench.WORKLOADS[name]
    print(f"{name}: kernel kind {kind}, {total} windows x {cfg['W']} bins ({cfg['n']} samples): windows differing from the oracle {nw}, bins {nb}, "
          f"worst deviation {worst:.2f} ulp of the window maximum{'' if first is None else ', first at window %d' % first}"
          f"   [gpu side {t_gpu:.0f} s, oracle {t_cpu:.0f} s on {cores} threads]", flush=True)
