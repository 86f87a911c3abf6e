"""Build the gfx950 shared library (quadrs_amd/libquadrs_hip.so) in-tree with hipcc.

-ffp-contract=off is load-bearing: the reference never fuses a*b+c on its f32 data path and
parity depends on separately rounded multiplies and adds.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# (source, extra flags)
SRC = [(os.path.join(HERE, "csrc", "quadrs_hip.hip"), [])]
DEPS = [s for s, _ in SRC] + [os.path.join(HERE, "csrc", f) for f in ("qd_chain.h", "qd_device.h", "qd_registry.h")] + [
    os.path.join(ROOT, "include", "quadrs_hip.h")]
OBJ_DIR = os.path.join(ROOT, "build", "obj")
OUT = os.path.join(HERE, "libquadrs_hip.so")

FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
         "-Rpass-analysis=kernel-resource-usage",
         # the tile queue's claim is ONE lane's atomic whose reply is read a phase later; LLVM's atomic optimizer would
         # rewrite it into a wave reduction + readfirstlane behind an immediate s_waitcnt vmcnt(0) (a full drain of the prefetch)
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
# per-kernel register / scratch figures of the last build (hipcc's kernel-resource-usage remarks), audited by
# tests/test_abi_cpu.py::test_builtin_kernels_do_not_spill: a scheduling accident that spills the FIR's products costs 8x
RESOURCES = os.path.join(ROOT, "build", "kernel_resources.json")


def parse_resource_remarks(text):
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return out


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


CLI_SRC = os.path.join(HERE, "cli", "quadrs_hip_cli.cpp")
CLI_OUT = os.path.join(HERE, "quadrs-hip")


def build_cli(force=False, verbose=False):
    """The C++ host driver (reference CLI grammar + operator chain over the C ABI)."""
    if not force and os.path.exists(CLI_OUT) and os.path.getmtime(CLI_OUT) >= max(
            os.path.getmtime(CLI_SRC), os.path.getmtime(OUT), os.path.getmtime(DEPS[-1])):
        return CLI_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI_OUT, CLI_SRC,
           "-L", HERE, "-lquadrs_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI_OUT


def build(force=False, verbose=False, extra=()):
    if force or needs_build():
        os.makedirs(OBJ_DIR, exist_ok=True)
        objs, procs = [], []
        for src, more in SRC:                      # the translation units compile in parallel
            obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
            cmd = [hipcc()] + FLAGS + more + list(extra) + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
            objs.append(obj)
        resources = {}
        for cmd, pr in procs:
            _, err = pr.communicate()
            resources.update(parse_resource_remarks(err))
            kept, skip = [], 0            # a remark is three lines (message, source echo, caret) behind optional "In file included" lines
            for l in err.splitlines():
                if "[-Rpass-analysis=kernel-resource-usage]" in l:
                    skip = 2
                    while kept and kept[-1].startswith("In file included from"):
                        kept.pop()
                    continue
                if skip:
                    skip -= 1
                    continue
                kept.append(l)
            diag = "\n".join(kept)
            if diag.strip():
                print(diag, file=sys.stderr)
            if pr.returncode != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        import json
        with open(RESOURCES, "w") as f:
            json.dump(resources, f, indent=0, sort_keys=True)
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lhiprtc", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    build_cli(force=force, verbose=verbose)
    return OUT


DEV_OUT = os.path.join(HERE, "libquadrs_hip_dev.so")


def build_dev(verbose=False, extra=()):
    """Development library (-DQD_DEVELOP): the timing-only ablation bits of the chain kernel and the QD_DEBUG_SKIP /
    QD_WG_PER_CU / QD_JIT_FLAGS / QD_JIT_NOSLP / QD_JIT_DUMP environment knobs exist only here.  The probe scripts under
    scripts/ load it through QD_LIB_PATH; tests, bench.py and the CLI use the shipped library."""
    obj_dir = os.path.join(ROOT, "build", "obj_dev")
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    for src, more in SRC:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        cmd = [hipcc()] + FLAGS + more + ["-DQD_DEVELOP"] + list(extra) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        objs.append(obj)
    for cmd, pr in procs:
        _, err = pr.communicate()
        if pr.returncode != 0:
            print(err, file=sys.stderr)
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DEV_OUT] + objs + ["-lhiprtc", "-ldl"])
    return DEV_OUT


if __name__ == "__main__":
    if "--dev" in sys.argv:
        print(build_dev(verbose=True, extra=["-DQD_STAMP"] if "--stamp" in sys.argv else ()))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(OUT)
