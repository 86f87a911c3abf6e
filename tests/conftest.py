import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the test matrix is run under QD_NO_FIXED / QD_JIT / QD_TUNE ...: the Python view translates them into qd_plan_options only
# under this gate (a user's environment never changes a plan: quadrs_amd/engine.py)
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns_gpu_ranks: starts rank processes that use the GPU; scheduled before any test that initialises it")
    # plan-time builds made by the tests go to a throw-away cache, not to ~/.cache/quadrs_hip
    if "QD_JIT_CACHE" not in os.environ:
        import atexit, shutil, tempfile
        d = tempfile.mkdtemp(prefix="quadrs_hip_jit_")
        os.environ["QD_JIT_CACHE"] = d
        atexit.register(shutil.rmtree, d, True)


def pytest_collection_modifyitems(config, items):
    """Tests that START OTHER PROCESSES which then use the GPU run first (marker `spawns_gpu_ranks`): the GPU boxes refuse an exec
    from a process that has already initialised the GPU (fork + exec of a rank is exactly that), so those tests must run
    while this pytest process has not touched the device yet.  Stable: everything else keeps its order."""
    first = [it for it in items if it.get_closest_marker("spawns_gpu_ranks")]
    if first:
        rest = [it for it in items if not it.get_closest_marker("spawns_gpu_ranks")]
        items[:] = first + rest


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def cupboard():
    with open(os.path.join(GOLDEN, "cupboard-superdec.sr400.cf32"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def fsk():
    with open(os.path.join(GOLDEN, "fsk-example-head65536.sr21M.cf32"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def engine():
    """The HIP engine; building is part of the fixture so a stale .so never passes silently."""
    from quadrs_amd import build as B
    B.build()
    import quadrs_amd
    return quadrs_amd
