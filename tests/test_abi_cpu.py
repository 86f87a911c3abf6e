"""CPU tests of the C-ABI library: it loads, exports every declared symbol, and its host-side
arithmetic (no GPU needed) agrees with the oracle.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(engine):
    from quadrs_amd import _ffi
    hdr = open(os.path.join(ROOT, "include", "quadrs_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(qd_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = C.CDLL(_ffi.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in include/quadrs_hip.h but not exported: {missing}"
    assert sorted(_ffi.SYMBOLS) == declared       # the Python binding covers the whole header


def test_struct_layout(engine):
    from quadrs_amd import _ffi
    assert C.sizeof(_ffi.ChainDesc) == 112 and C.sizeof(_ffi.PlanInfo) == 80


def test_host_arithmetic_matches_oracle(engine, oracle):
    L = engine._ffi.lib()
    for fmt in range(4):
        assert L.qd_pair_bytes(fmt) == oracle.lib().qo_pair_bytes(fmt)
    for f, sr in ((280000, 21_000_000), (-280000, 21_000_000), (1, 3), (12_345_678, 100_000_000)):
        assert engine.shift_ratio(f, sr) == oracle.shift_ratio(f, sr)
    for T, fc, sr in ((40, 2_000_000, 21_000_000), (400, 200_000, 21_000_000), (512, 5_000_000, 100_000_000), (2, 1, 7)):
        assert bits_equal(engine.lowpass_design(fc, sr, T), oracle.taps(fc, sr, T))
    # odd size: sinc(0) = NaN at the centre tap, as in the reference
    a, b = engine.lowpass_design(1000, 48000, 5), oracle.taps(1000, 48000, 5)
    assert np.isnan(a).all() and np.isnan(b).all()     # NaN sum poisons every tap after normalisation


@pytest.mark.parametrize("kwargs,code", [
    (dict(width=100), 2),                                   # Radix4 needs a power of two -> panic
    (dict(width=128, stride=0), 1),                         # never terminates -> invalid
    (dict(width=128, shift_hz=10_500_000), 2),              # |f| < sr/2 (src/shift.rs:20-23)
    (dict(width=128, lowpass=(1000, 0, 40)), 2),            # decimate 0
    (dict(width=128, lowpass=(1000, 8, 1)), 2),             # size < 2
    (dict(width=4096, n_samples=100), 2),                   # len < width underflow (src/fft.rs:28)
    (dict(width=128, lowpass=(1000, 8, 40), n_samples=30), 2),   # inner.len() < filter.len()
])
def test_plan_validation_precedes_any_gpu_call(engine, kwargs, code):
    """Argument checks mirror the reference's asserts and run before the first HIP call."""
    kw = dict(fmt=engine.FMT_CF32, sample_rate=21_000_000, n_samples=1 << 20)
    kw.update(kwargs)
    with pytest.raises(engine.QuadrsError) as ei:
        engine.Plan(**kw)
    assert ei.value.code == code


def test_missing_library_fails_loudly(engine, monkeypatch):
    from quadrs_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libquadrs_hip.so")
    with pytest.raises(ImportError):
        _ffi.lib()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under quadrs_amd/ or include/ may reference it."""
    bad = []
    for d in ("quadrs_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"quadrs_oracle|from oracle|import oracle|qo_[a-z]", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_options_struct_and_validation(engine):
    """qd_plan_options is checked before any GPU call; sizes match the header's layout."""
    from quadrs_amd import _ffi
    assert C.sizeof(_ffi.PlanOptions) == 4 * 4 + 8 + 4 + 16 * 4 + 8 * 4 + 4 and C.sizeof(_ffi.ShardInfo) == 48 and C.sizeof(_ffi.PlanStats) == 40
    kw = dict(fmt=engine.FMT_CF32, sample_rate=21_000_000, n_samples=1 << 20, width=128)
    for bad in (dict(kernel_policy=9), dict(nco_order=3), dict(copy_threads=1000), dict(chunk_bytes=10)):
        with pytest.raises(engine.QuadrsError) as ei:
            engine.Plan(**kw, **bad)
        assert ei.value.code == 1, bad
    o = engine.plan_options()
    o.struct_size = 12
    with pytest.raises(engine.QuadrsError) as ei:
        engine.Plan(**kw, options=o)
    assert ei.value.code == 1


def test_shipped_library_reads_no_tuning_environment(engine):
    """'the C ABI takes an explicit struct, no env vars'.  The only getenv calls allowed in the product
    source are the development-build helper (compiled to `return nullptr` without -DQD_DEVELOP) and the location of the
    on-disk code-object cache; the ablation bits are compiled out of the kernels unless QD_DEVELOP is defined."""
    src = open(os.path.join(ROOT, "quadrs_amd", "csrc", "quadrs_hip.hip")).read()
    calls = [m.start() for m in re.finditer(r"\bgetenv\(", src)]
    allowed = []
    for pos in calls:
        line = src[src.rfind("\n", 0, pos) + 1:src.find("\n", pos)]
        allowed.append(("dev_env(const char *name)" in line) or any(k in line for k in ('"QD_JIT_CACHE"', '"XDG_CACHE_HOME"', '"HOME"')))
    assert calls and all(allowed), [src[p - 40:p + 40] for p, ok in zip(calls, allowed) if not ok]
    chain = open(os.path.join(ROOT, "quadrs_amd", "csrc", "qd_chain.h")).read()
    assert "#ifdef QD_DEVELOP" in chain and not re.search(r"P\.dbg\s*&", chain)
    import subprocess
    strings = subprocess.run(["strings", "-a", engine._ffi.LIB_PATH], capture_output=True, text=True).stdout
    for knob in ("QD_DEBUG_SKIP", "QD_TUNE", "QD_JIT_FLAGS", "QD_WG_PER_CU", "QD_NO_FIXED", "QD_CHUNK_MB"):
        assert knob not in strings, knob


def test_builtin_kernels_do_not_spill(engine):
    """hipcc's per-kernel resource remarks of the build that produced the library (build/kernel_resources.json): the
    shape-specialised chain kernels keep their arithmetic in registers.  A scheduling accident here is silent and ruinous —
    the 200-tap kernel once spilled every FIR product (185 VGPRs of scratch) and ran 8.7x slower with parity still green."""
    import json
    from quadrs_amd import build as B
    if not os.path.exists(B.RESOURCES) or os.path.getmtime(B.RESOURCES) < os.path.getmtime(B.OUT) - 600:
        B.build(force=True)
    res = json.load(open(B.RESOURCES))
    fixed = {k: v for k, v in res.items() if "k_chain" in k and "FixedGeo" in k}
    assert len(fixed) >= 8, sorted(res)
    bad = {k: v for k, v in fixed.items() if v.get("VGPRs Spill", 0) > 2 or v.get("ScratchSize", 0) > 16}
    assert not bad, bad
    # SGPR spills are v_writelane / v_readlane on the vector unit.  The 128-point kernels (cfg2, cfg3') once carried 79-85 of them
    # — the tile queue's help-the-others loop, unrolled and hoisted out of the tile loop — and now 10-16; the long-filter kernels
    # (1024 threads, unrolled 400 / 512-tap FIRs) keep more.
    for k, v in fixed.items():
        short = "Lj128ELj128E" in k
        assert v.get("SGPRs Spill", 0) <= (24 if short else 160), (k, v)
    # round 4: EVERY kernel of the library — the runtime-geometry (DynGeo) chain kernels and the wave-local kernels included — without
    # scratch.  The generic kernels serve every stream below 1 GiB that has no cached plan-time build (both README examples); with a shift
    # they are budgeted for three waves per SIMD instead of spilling up to 74 VGPRs at four (round 3).  SGPR spills (lanes of a VGPR, no
    # memory) are bounded at today's figures: the generic kernels keep ~50 runtime geometry values uniform.
    generic = {k: v for k, v in res.items() if "DynGeo" in k}
    assert len(generic) >= 60, len(generic)
    bad = {k: v for k, v in res.items() if v.get("ScratchSize", 0) > 0 or v.get("VGPRs Spill", 0) > 0}
    bad = {k: v for k, v in bad.items() if not ("FixedGeo" in k and v.get("VGPRs Spill", 0) <= 2 and v.get("ScratchSize", 0) <= 16)}       # (the bound above)
    assert not bad, bad
    assert all(v.get("SGPRs Spill", 0) <= 160 for v in generic.values()), {k: v for k, v in generic.items() if v.get("SGPRs Spill", 0) > 160}
