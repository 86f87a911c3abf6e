"""Build the gfx950 shared library (quadrs_amd/libquadrs_hip.so) in-tree with hipcc.

-ffp-contract=off is load-bearing: the reference never fuses a*b+c on its f32 data path and
parity depends on separately rounded multiplies and adds.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "quadrs_hip.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", f) for f in ("qd_chain.h", "qd_device.h")] + [
    os.path.join(ROOT, "include", "quadrs_hip.h")]
OUT = os.path.join(HERE, "libquadrs_hip.so")

FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
