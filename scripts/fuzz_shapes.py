"""Development: random chain shapes, plan-time specialised kernels vs the generic kernel, bit for bit
(the same generator as tests/test_gpu_robustness.py::test_random_shapes_specialised_equals_generic, more shapes).
usage: fuzz_shapes.py [n_shapes] [seed] [variants|auto]   (third argument "variants": only geometries the kernel variant flags apply to, each with a
variant tiling; "auto": the families the plan-time variant selection serves, no hint — the library chooses)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quadrs_amd as Q
from util import fuzz_chain_shapes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else ""
stats = []
checked, bad = fuzz_chain_shapes(Q, n, seed, log=lambda m: print(m, flush=True), variants_only=mode == "variants" or (mode and mode != "auto"), auto_only=mode == "auto", stats=stats)
if mode == "auto":
    print(f"plans with variant flags: {sum(1 for k, f in stats if k == 2 and f)} of {checked}")
print(f"checked {checked} shapes, mismatching: {len(bad)}")
sys.exit(1 if bad else 0)
