import numpy as np

import os

# /root/reference/README.md:167, copied verbatim as an expected-output fixture
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_ook_line167.txt")) as _f:
    README_OOK = _f.read().strip()


def ook_pipeline(text):
    """The README's sed/tr pipeline (README.md:122-167) over spark_fft's stdout bytes."""
    import re
    bits = []
    for line in text.decode("utf-8").split("\n")[:-1]:
        line = re.sub(r"^.    .$", ".", line)
        line = re.sub(r"....*", "X", line)
        bits.append(line)
    s = "".join(bits).replace(".", "o")     # the README writes '.' in one step and 'o' in the next
    return re.sub(r"o{5,10}", "B", re.sub(r"X{6,10}", "A", s))


def ulp_diff(a, b):
    """distance in units of f32 representable steps (sign-magnitude ordered)"""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def ulp_of(x):
    """spacing of f32 at |x|"""
    x = np.abs(np.asarray(x, dtype=np.float32))
    return np.spacing(np.maximum(x, np.float32(1e-45)))


def complex_ulp_err(ref, got):
    """|got - ref| per component in units of ulp(max(|re|,|im|)) of the reference sample —
    the cf32 tolerance unit used throughout (north_star: 'within 1 ulp on cf32')."""
    ref = np.asarray(ref, dtype=np.float32).reshape(-1, 2)
    got = np.asarray(got, dtype=np.float32).reshape(-1, 2)
    scale = ulp_of(np.max(np.abs(ref), axis=1)).astype(np.float64)
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max(axis=1)
    return d / scale


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def fuzz_chain_shapes(Q, n_shapes, seed, log=None, oracle=None, variants_only=False, auto_only=False, stats=None):
    """Random chain shapes through the plan-time compiler (QD_JIT=1; a quarter of them with a QD_TUNE tiling that
    exercises the register-tiled FIR / 16-byte LDS rows / wide workgroups) against the generic kernel (QD_JIT=0),
    bit for bit; with `oracle` (tests only) the first windows are also checked against the CPU oracle: bit-exact without
    a shift stage, within 1 ulp of the window maximum with one.  Returns (checked, mismatching descriptions).
    Q is the quadrs_amd package."""
    import os
    rng = np.random.default_rng(seed)
    checked, bad = 0, []
    saved = {k: os.environ.get(k) for k in ("QD_TUNE", "QD_JIT")}
    try:
        for _ in range(n_shapes):
            fmt = int(rng.integers(0, 4))
            W = 1 << int(rng.integers(2, 11))
            S = int(rng.choice([W, W, max(1, W // 2), max(1, W // 4), int(rng.integers(1, 2 * W + 1))]))
            D = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 32, 64]))
            T = int(rng.choice([2, 8, 9, 16, 40, 48, 64, 100, 128, 200, 256, 400, 512, 800]))
            if auto_only:                           # the families the plan-time variant selection serves, NO hint: the library chooses
                W = int(rng.choice([64, 128, 256, 512, 1024]))
                S = W
                D = int(rng.choice([4, 8, 16, 32]))
                T = int(rng.choice([64, 72, 96, 128, 160, 192, 200, 256, 384, 400, 512]))
                fmt = int(rng.choice([0, 0, 1, 3]))
                if rng.random() < 0.3:                 # overlapping windows with a long filter: the three-stage kernel's family
                    W = int(rng.choice([64, 128]))
                    S = int(rng.choice([16, W // 4, W // 2]))
                    D = int(rng.choice([8, 16, 32]))
                    T = int(rng.choice([t for t in (128, 200, 256, 400, 512) if t >= 8 * D]))
            if variants_only:                       # geometries the FLAGS_ variants apply to, every shape with a variant tiling
                D = int(rng.choice([8, 16, 32, 64]))
                T = int(rng.choice([32, 40, 48, 64, 96, 128, 200, 256, 400, 512, 800]))
                if rng.random() < 0.5:
                    S = W
            if (W * D + T) * 8 * 1.2 > 150 * 1024:
                continue
            shift = None if rng.random() < 0.25 else int(rng.integers(-3_000_000, 3_000_000))
            N = (int(rng.integers(3, 400)) * S + W) * D + T + int(rng.integers(0, D + 1))
            bps = {0: 8, 1: 2, 2: 2, 3: 4}[fmt]
            data = rng.integers(0, 256, N * bps, dtype=np.uint8)
            if fmt == 0:
                data = (rng.standard_normal((N, 2)).astype(np.float32) * 0.05).view(np.uint8).reshape(-1)
            tune = None
            if auto_only:
                pass
            elif not variants_only and rng.random() < 0.25 and D % 8 == 0 and T % 16 == 0:
                tune = "%d:%d:%d:8:%d:%d" % (rng.integers(1, 4), rng.choice([256, 512, 1024]), rng.choice([1, 2]), rng.choice([2, 4]),
                                            rng.choice([1, 2]))
            elif (variants_only or rng.random() < 0.35) and D % 8 == 0 and T % 8 == 0 and T >= 32:
                # the FLAGS_ variants of the built-in kernels on shapes of the fuzzer's choosing: packed pair FIR (4), row-aligned
                # phase 1 (8), deferred FFT (64, two slots), straight-line shared FIR (32), packed two-output tile (128)
                flags, batch = [(4, 1), (12, 1), (68, 2), (76, 2), (32, 1), (128, 1), (192, 2), (4, 2)][int(rng.integers(0, 8))]
                tune = "%d:%d:%d:4:4:2:%d:0" % (rng.integers(1, 3), rng.choice([256, 512, 1024]), 2 if flags & 128 else 1, batch | (flags << 8))
            epi = int(rng.choice([0, 0, 1, 2]))             # f32 norms / glyph codes / bucket digits
            outs, info = {}, {}
            for mode in ("0", "1"):
                os.environ.pop("QD_TUNE", None)
                os.environ["QD_JIT"] = mode
                if mode == "1" and tune:
                    os.environ["QD_TUNE"] = tune
                try:
                    try:
                        p = Q.Plan(fmt, 21_000_000, N, shift_hz=shift, lowpass=(1_000_000, D, T), width=W, stride=S, epilogue=epi)
                    except Q.QuadrsError as e:
                        if "tile_hint" not in str(e) or "QD_TUNE" not in os.environ:
                            raise
                        os.environ.pop("QD_TUNE")            # a tiling this shape cannot take (LDS, geometry): the library's own choice
                        tune = None
                        p = Q.Plan(fmt, 21_000_000, N, shift_hz=shift, lowpass=(1_000_000, D, T), width=W, stride=S, epilogue=epi)
                    outs[mode] = p.run_host(data)
                    info[mode] = (p.info.kernel_kind, p.info.tile_windows, p.info.threads, p.info.kernel_flags)
                    if stats is not None and mode == "1":
                        stats.append((int(p.info.kernel_kind), int(p.info.kernel_flags)))
                except Q.QuadrsError as e:
                    outs[mode] = str(e)
            a, b = outs["0"], outs["1"]
            ok = (isinstance(a, str) and isinstance(b, str)) or (
                not isinstance(a, str) and not isinstance(b, str) and a.shape == b.shape and a.tobytes() == b.tobytes())
            if ok and epi == 0 and oracle is not None and not isinstance(a, str) and a.shape[0] > 0:
                ch = oracle.Chain.from_bytes(data.tobytes(), fmt, 21_000_000)
                if shift is not None:
                    ch = ch.shift(shift)
                ref, _ = ch.lowpass(1_000_000, D, T).spark_fft(W, S, max_windows=24)
                got = a[:ref.shape[0]]
                if shift is None:
                    ok = bits_equal(ref, got)
                else:
                    # odd tap counts put a 0/0 in the middle of the reference's windowed sinc: NaN spectra, on both sides
                    both_nan = np.isnan(ref) & np.isnan(got)
                    with np.errstate(invalid="ignore"):
                        scale = ulp_of(np.nanmax(np.where(np.isnan(ref), -np.inf, ref), axis=-1, keepdims=True)).astype(np.float64)
                        close = np.abs(ref.astype(np.float64) - got.astype(np.float64)) <= 1.0 * scale
                    ok = bool((close | both_nan).all())
            desc = f"fmt={fmt} W={W} S={S} D={D} T={T} shift={shift} N={N} epi={epi} tune={tune} kinds={info}"
            if log:
                log(("ok  " if ok else "BAD ") + desc)
            checked += 1
            if not ok:
                bad.append(desc)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return checked, bad


def fuzz_nofir_shapes(Q, n_shapes, seed, log=None, oracle=None, stats=None):
    """Random chains WITHOUT a lowpass (`from F [shift] sparkfft -width W -stride S`): the plan-time builds of the wave-local kernel family
    (k_spark swizzled, k_spark2, k_spark0, interleaved launches; QD_KERNEL_SPECIALISE) against the generic chain kernel, bit for bit
    without a shift and where the NCO row grids agree (cf32), within the NCO's tolerance otherwise — whole streams and a random window
    sub-range each; with `oracle` (tests only) also against the CPU oracle: bit-exact without a shift, 1 ulp of the window maximum with.
    Returns (checked, mismatching descriptions)."""
    from quadrs_amd import _ffi
    rng = np.random.default_rng(seed)
    checked, bad = 0, []
    for _ in range(n_shapes):
        fmt = int(rng.integers(0, 4))
        W = 1 << int(rng.integers(0, 11))
        S = int(rng.choice([W, W, max(1, W // 2), max(1, W // 4), int(rng.integers(1, W + 1)), int(rng.integers(1, 2 * W + 1))]))
        shift = None if rng.random() < 0.55 else int(rng.integers(-3_000_000, 3_000_000))
        epi = int(rng.choice([0, 0, 1, 2]))
        if epi == 2 and W < 2:
            epi = 0
        bps = {0: 8, 1: 2, 2: 2, 3: 4}[fmt]
        N = int(rng.integers(2, 600)) * S + W + int(rng.integers(0, max(2, S)))
        N = min(N, 300_000)
        data = rng.integers(0, 256, N * bps, dtype=np.uint8)
        if fmt == 0:
            data = (rng.standard_normal((N, 2)).astype(np.float32) * 0.05).view(np.uint8).reshape(-1)
        raw = data.tobytes()
        kw = dict(shift_hz=shift, width=W, stride=S, epilogue=epi, rng=(0.01, 0.5) if fmt == 0 else (0.3, 30.0))
        desc = f"fmt={fmt} W={W} S={S} shift={shift} N={N} epi={epi}"
        try:
            j = Q.Plan(fmt, 21_000_000, N, kernel_policy=_ffi.KERNEL_SPECIALISE, **kw)
            g = Q.Plan(fmt, 21_000_000, N, kernel_policy=_ffi.KERNEL_GENERIC, **kw)
        except Q.QuadrsError as e:
            if log:
                log("skip " + desc + ": " + str(e)[:80])
            continue
        a, b = j.run_host(raw), g.run_host(raw)
        desc += f" kind={j.info.kernel_kind} flags={j.info.kernel_flags}"
        if stats is not None:
            stats.append((int(j.info.kernel_kind), int(j.info.kernel_flags)))

        # bit for bit where the two kernels tile the NCO alike (no shift; cf32 with windows side by side); within the NCO's tolerance where
        # they do not (8-bit formats and cs16: rows of 512 against 1024 samples; overlapping windows behind a shift: a row grid per
        # interleaved launch) — DESIGN section 4.  Sinks that quantise then differ in a cell next to a threshold at most.
        exact = shift is None or (fmt == 0 and S >= W)

        def same(x, y):
            if x.shape != y.shape:
                return False
            if exact:
                return x.tobytes() == y.tobytes()
            if x.dtype != np.float32:
                return float((x != y).mean()) <= 5e-3
            scale = ulp_of(np.maximum(np.abs(y).max(axis=-1, keepdims=True), 1e-30)).astype(np.float64)
            return bool((np.abs(x.astype(np.float64) - y.astype(np.float64)) <= scale).all())
        ok = same(a, b)
        nw = j.n_windows
        if ok and nw > 2:
            w0 = int(rng.integers(1, nw))
            cnt = int(rng.integers(1, nw - w0 + 1))
            first, count = j.src_range(w0, cnt)
            sub = j.run_host(raw[first * bps:(first + count) * bps], w0, cnt, src_first=first)
            ok = same(sub, a[w0:w0 + cnt])
            if not ok:
                desc += f" SUB-RANGE w0={w0} cnt={cnt}"
        if ok and epi == 0 and oracle is not None and nw > 0:
            ch = oracle.Chain.from_bytes(raw, fmt, 21_000_000)
            if shift is not None:
                ch = ch.shift(shift)
            ref, _ = ch.spark_fft(W, S, max_windows=32)
            got = a[:ref.shape[0]]
            if shift is None:
                ok = bits_equal(ref, got)
            else:
                scale = ulp_of(np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-30)).astype(np.float64)
                ok = bool((np.abs(ref.astype(np.float64) - got.astype(np.float64)) <= scale).all())
            if not ok:
                desc += " ORACLE"
        j.close(); g.close()
        if log:
            log(("ok  " if ok else "BAD ") + desc)
        checked += 1
        if not ok:
            bad.append(desc)
    return checked, bad


def full_size_census(Q, O, bench, name):
    """EVERY window of BASELINE workload `name` at its full size: HIP chain kernel against the CPU oracle in its cheapest exact
    form (FIR at the decimated positions only, same products, same order) on all host cores, same input bytes, absolute sample
    indices.  Returns (n_windows, kernel_kind, windows_differing, bins_differing, worst deviation in ulp of the window maximum,
    first differing window or None, seconds on the GPU side, seconds in the oracle, threads)."""
    import concurrent.futures as cf
    import os
    import time
    import torch
    dev = torch.device("cuda", 0)
    cores = max(1, min(len(os.sched_getaffinity(0)), 32))
    cfg = bench.WORKLOADS[name]
    t0 = time.perf_counter()
    if name == "cfg4":
        src = torch.empty(cfg["n"], 2, dtype=torch.float32, device=dev)
        tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
        for a in range(0, cfg["n"], 1 << 28):
            Q.gen_device(tones, cfg["sr"], a, src[a:a + (1 << 28)])
    else:
        src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
    p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    out = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    host = src.view(torch.uint8).reshape(-1).cpu().numpy()
    del src, out
    torch.cuda.empty_cache()
    t_gpu = time.perf_counter() - t0
    # the oracle on the very same bytes; no copy of the slab (the C side reads through the pointer)
    ch = O.Chain()
    ch._node = O.lib().qo_source_mem(O._p(host), host.size, cfg["fmt"], cfg["sr"])
    ch._keep.append(host)
    if cfg["shift"] is not None:
        ch = ch.shift(cfg["shift"])
    ch = ch.lowpass(*cfg["lp"])
    total = O.lib().qo_spark_window_count(ch.len(), cfg["W"], cfg["S"])
    assert total == p.n_windows, (total, p.n_windows)
    piece = 8192
    jobs = [(a, min(piece, total - a)) for a in range(0, total, piece)]

    def work(job):
        a, n = job
        ref, _ = ch.spark_fft(cfg["W"], cfg["S"], first_window=a, max_windows=n, want_codes=False)
        g = got[a:a + n]
        ne = ref.view(np.uint32) != g.view(np.uint32)
        if not ne.any():
            return 0, 0, 0.0, None
        scale = np.spacing(np.abs(ref).max(axis=1, keepdims=True).astype(np.float32)).astype(np.float64)
        err = (np.abs(ref.astype(np.float64) - g.astype(np.float64)) / scale).max()
        rows = np.nonzero(ne.any(axis=1))[0]
        return len(rows), int(ne.sum()), float(err), int(a + rows[0])

    nw = nb = 0
    worst, first = 0.0, None
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        for a, b, err, f in ex.map(work, jobs):
            nw += a; nb += b; worst = max(worst, err)
            if f is not None and first is None:
                first = f
    return total, int(p.info.kernel_kind), nw, nb, worst, first, t_gpu, time.perf_counter() - t0, cores
