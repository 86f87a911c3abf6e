"""Development: random lowpass-free chains, plan-time builds of the wave-local kernels vs the generic kernel and the CPU oracle
(the generator of tests/test_gpu_robustness.py::test_random_lowpass_free_shapes, more shapes).  usage: fuzz_nofir.py [n_shapes] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quadrs_amd as Q
from util import fuzz_nofir_shapes
from oracle import oracle as O        # the checker (test infrastructure), as in tests/conftest.py
O.lib()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stats = []
checked, bad = fuzz_nofir_shapes(Q, n, seed, log=lambda m: print(m, flush=True), oracle=O, stats=stats)
print(f"checked {checked} shapes, on the wave-local family: {sum(1 for k, f in stats if f & 524288)}, mismatching: {len(bad)}")
sys.exit(1 if bad else 0)
