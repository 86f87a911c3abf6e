"""Development: random chain shapes, plan-time specialised kernels vs the generic kernel, bit for bit
(the same generator as tests/test_gpu_robustness.py::test_random_shapes_specialised_equals_generic, more shapes).
usage: fuzz_shapes.py [n_shapes] [seed] [variants]   (third argument: only geometries the kernel variant flags apply to, each with a variant tiling)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quadrs_amd as Q
from util import fuzz_chain_shapes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
checked, bad = fuzz_chain_shapes(Q, n, seed, log=lambda m: print(m, flush=True), variants_only=len(sys.argv) > 3)
print(f"checked {checked} shapes, mismatching: {len(bad)}")
sys.exit(1 if bad else 0)
