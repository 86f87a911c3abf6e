"""GPU robustness: far-offset slabs (second-order NCO kernels), device-memory fine-grained calls,
concurrent use of one plan, degenerate sizes."""
import ctypes as C
import threading

import os

import numpy as np
import pytest

from util import bits_equal, complex_ulp_err, ulp_of

pytestmark = pytest.mark.gpu


def _norms_close(ref, got):
    exact = (ref.view(np.uint32) == got.view(np.uint32)).mean()
    scale = ulp_of(ref.max(axis=-1, keepdims=True)).astype(np.float64)
    worst = (np.abs(ref.astype(np.float64) - got.astype(np.float64)) / scale).max()
    return exact, worst


@pytest.mark.parametrize("lp,W,S", [((2_000_000, 16, 40), 128, 128),     # cfg2 shape  -> FixedGeo, NCO order 2
                                    ((200_000, 32, 200), 128, 128),      # cfg3' shape -> FixedGeo, NCO order 2
                                    ((200_000, 32, 400), 64, 16),        # README FSK shape (shared FIR), order 2
                                    ((700_000, 10, 24), 16, 5)])         # generic kernel, order 2
def test_far_offset_slab_uses_second_order_nco(engine, oracle, lp, W, S):
    """A slab deep inside a 2^34-sample stream (cfg5 territory): |place| ~ 7e8 rad, where the NCO needs
    its second-order term; every window is checked against the oracle evaluated at absolute indices."""
    fc, D, T = lp
    N = 1 << 34
    p = engine.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W, stride=S)
    w0 = (1 << 33) // (S * D) + 12345
    nwin = 24
    first, count = p.src_range(w0, nwin)
    rng = np.random.default_rng(W * 7 + S)
    x = (rng.standard_normal((count, 2)) * 0.03).astype(np.float32)
    got = p.run_host(x.tobytes(), w0, nwin, src_first=first)
    ratio = oracle.shift_ratio(280000, 21_000_000)
    taps = oracle.taps(fc, 21_000_000, T)
    ref = np.empty_like(got)
    for i in range(nwin):
        a = (w0 + i) * S * D - first
        raw = oracle.shift_apply(x[a:a + W * D + T], (w0 + i) * S * D, ratio)
        k, dec = oracle.lowpass_block(taps, D, raw)
        assert k == W
        y = oracle.fft(dec)
        ref[i] = oracle.norm(y)[np.r_[W // 2:W, 0:W // 2]]
    exact, worst = _norms_close(ref, got)
    from test_gpu_parity import record_observed
    record_observed(f"deep-stream windows W={W} S={S}", exact_fraction=float(exact), worst_ulp_of_window_max=float(worst))
    assert exact >= 0.9999 and worst <= 1.0, (exact, worst)


def test_fine_grained_calls_on_device_memory(engine, oracle):
    import torch
    from quadrs_amd import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(3)
    # unpack
    raw8 = rng.integers(0, 256, 2 * 5000, dtype=np.uint8)
    d_in = torch.from_numpy(raw8).cuda()
    d_out = torch.empty(5000, 2, dtype=torch.float32, device="cuda")
    _ffi.check(L.qd_unpack(_ffi.FMT_CS8, C.c_void_p(d_in.data_ptr()), 5000, C.c_void_p(d_out.data_ptr()), _ffi.MEM_DEVICE))
    torch.cuda.synchronize()
    assert bits_equal(d_out.cpu().numpy(), oracle.unpack(oracle.FMT_CS8, raw8.tobytes()))
    # shift in place
    x = (rng.standard_normal((30_000, 2)) * 0.1).astype(np.float32)
    ratio = engine.shift_ratio(-123_456, 2_000_000)
    d_x = torch.from_numpy(x).cuda()
    _ffi.check(L.qd_shift(C.c_void_p(d_x.data_ptr()), 30_000, 777_777_777, ratio, _ffi.MEM_DEVICE))
    assert complex_ulp_err(oracle.shift_apply(x, 777_777_777, ratio), d_x.cpu().numpy()).max() <= 1.0
    # lowpass block
    taps = oracle.taps(1000, 16000, 64)
    raw = rng.standard_normal((200 * 8 + 64, 2)).astype(np.float32)
    d_raw = torch.from_numpy(raw).cuda()
    d_dec = torch.empty(200, 2, dtype=torch.float32, device="cuda")
    produced = C.c_size_t(0)
    _ffi.check(L.qd_lowpass_block(taps.ctypes.data_as(C.c_void_p), 64, 8, C.c_void_p(d_raw.data_ptr()), raw.shape[0],
                                  C.c_void_p(d_dec.data_ptr()), 200, C.byref(produced), _ffi.MEM_DEVICE))
    assert produced.value == 200 and bits_equal(d_dec.cpu().numpy(), oracle.lowpass_block(taps, 8, raw)[1])
    # fft batch
    xin = rng.standard_normal((256 * 4, 2)).astype(np.float32)
    d_xin = torch.from_numpy(xin).cuda()
    d_n = torch.empty(4, 256, dtype=torch.float32, device="cuda")
    _ffi.check(L.qd_fft_norm_batch(C.c_void_p(d_xin.data_ptr()), 256, 4, 256, C.c_void_p(d_n.data_ptr()), _ffi.MEM_DEVICE))
    for i in range(4):
        assert bits_equal(d_n[i].cpu().numpy(), oracle.norm(oracle.fft(xin[i * 256:(i + 1) * 256]))[np.r_[128:256, 0:128]])
    # gen
    cos = np.array([1000, -2500], dtype=np.int64)
    d_g = torch.empty(100, 2, dtype=torch.float32, device="cuda")
    _ffi.check(L.qd_gen(cos.ctypes.data_as(C.c_void_p), 2, 48000, 1 << 20, 100, C.c_void_p(d_g.data_ptr()), _ffi.MEM_DEVICE))
    assert complex_ulp_err(oracle.Chain.gen([1000, -2500], 48000).read_at(1 << 20, 100)[1], d_g.cpu().numpy()).max() <= 1.0


def test_concurrent_runs_on_one_plan(engine):
    """Samples: Sync + Send — concurrent calls on one object are legal; the plan serialises them."""
    rng = np.random.default_rng(11)
    N = 400_000
    x = (rng.standard_normal((N, 2)) * 0.05).astype(np.float32).tobytes()
    p = engine.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128)
    want = p.run_host(x)
    results, errors = [None] * 4, []

    def work(i):
        try:
            results[i] = p.run_host(x)
        except Exception as e:       # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors and all(bits_equal(r, want) for r in results)


def test_caller_streams_may_be_destroyed_between_calls(engine):
    """The library remembers the last stream of a launch context / workspace only to COMPARE it; ordering across streams is an event.
    A caller that destroys its stream after a call and comes back on a new one (a table rewrite, a workspace regrow in between)
    must get the same bits — and no use of the dead handle."""
    import torch
    import bench
    from quadrs_amd import _ffi
    dev = torch.device("cuda", 0)
    hip = C.CDLL("libamdhip64.so")
    L = _ffi.lib()

    def new_stream():
        h = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(h)) == 0
        return h

    cfg = bench.WORKLOADS["cfg2"]
    n = 1 << 22
    src = bench.synth_slab(torch, 0, 0, n, 0x5EED0002, dev)
    p = engine.Plan(0, cfg["sr"], n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    want = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, want)
    torch.cuda.synchronize()
    half = (p.n_windows // 2) & ~1
    for rnd in range(3):
        st = new_stream()
        got = torch.zeros_like(want)
        # two sub-ranges: the second one makes the launch context rewrite its row table (other rows) on the new stream
        p.run_device(src, got[:half], first_window=0, n_windows=half, stream=st.value)
        assert hip.hipStreamSynchronize(st) == 0
        assert hip.hipStreamDestroy(st) == 0
        st2 = new_stream()
        p.run_device(src, got[half:], first_window=half, n_windows=p.n_windows - half, stream=st2.value)
        assert hip.hipStreamSynchronize(st2) == 0
        assert hip.hipStreamDestroy(st2) == 0
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), rnd
    # fine-grained calls: workspaces change hands between streams that no longer exist; growing sizes force a regrow
    x = torch.randn(8192 * 64, 2, device=dev)
    for k, rows in enumerate((512, 2048, 8192)):
        ref = torch.empty(rows, 64, device=dev)
        _ffi.check(L.qd_set_stream(None))
        _ffi.check(L.qd_fft_norm_batch(C.c_void_p(x.data_ptr()), 64, rows, 64, C.c_void_p(ref.data_ptr()), _ffi.MEM_DEVICE))
        torch.cuda.synchronize()
        st = new_stream()
        _ffi.check(L.qd_set_stream(st))
        host_in = x[: rows * 64].cpu().numpy()
        host_out = np.zeros((rows, 64), dtype=np.float32)
        _ffi.check(L.qd_fft_norm_batch(host_in.ctypes.data_as(C.c_void_p), 64, rows, 64, host_out.ctypes.data_as(C.c_void_p), _ffi.MEM_HOST))
        _ffi.check(L.qd_set_stream(None))
        assert hip.hipStreamDestroy(st) == 0
        assert np.array_equal(host_out.view(np.int32), ref.cpu().numpy().view(np.int32)), k


def test_two_streams_on_one_plan_device_path(engine):
    """every device-path launch of a plan shares one set of tile-queue counters and row tables.  Two launches of ONE
    plan on DIFFERENT streams, each big enough for the dynamic tile queue (>= 4 tiles per workgroup of the grid), must not
    claim tiles from the same counters at the same time: the library orders launches of one launch context across streams
    (an event wait on the device).  Bit for bit against a single-stream run, several rounds."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    n = 1 << 27                                            # cfg2's shape at full size: 65 535 tiles of two windows >> 4 * 1024
    cfg = bench.WORKLOADS["cfg2"]
    src = bench.synth_slab(torch, 0, 0, n, 0x5EED0002, dev)
    p = engine.Plan(0, cfg["sr"], n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    want = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, want)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros_like(want)
    b = torch.zeros_like(want)
    for _ in range(4):
        a.zero_(); b.zero_()
        torch.cuda.synchronize()
        p.run_device(src, a, stream=s1.cuda_stream)
        p.run_device(src, b, stream=s2.cuda_stream)
        p.run_device(src, a, stream=s1.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), want.view(torch.int32)) and torch.equal(b.view(torch.int32), want.view(torch.int32))
    # the fine-grained FFT call shares one cached plan between threads with different qd_set_stream streams
    x = torch.randn(4096 * 128, 2, device=dev)
    ref = torch.empty(4096, 128, device=dev)
    import ctypes as C
    from quadrs_amd import _ffi
    L = _ffi.lib()
    _ffi.check(L.qd_fft_norm_batch(C.c_void_p(x.data_ptr()), 128, 4096, 128, C.c_void_p(ref.data_ptr()), _ffi.MEM_DEVICE))
    torch.cuda.synchronize()
    outs = [torch.zeros_like(ref) for _ in range(2)]
    errs = []

    def work(i, stream):
        try:
            _ffi.check(L.qd_set_stream(C.c_void_p(stream.cuda_stream)))
            for _ in range(8):
                _ffi.check(L.qd_fft_norm_batch(C.c_void_p(x.data_ptr()), 128, 4096, 128, C.c_void_p(outs[i].data_ptr()), _ffi.MEM_DEVICE))
            _ffi.check(L.qd_set_stream(None))
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i, st)) for i, st in enumerate((s1, s2))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errs and all(torch.equal(o.view(torch.int32), ref.view(torch.int32)) for o in outs)


def test_degenerate_sizes(engine, oracle):
    # exactly one admissible window minus one: len == W  ->  the strict `<` loop runs zero times
    x = np.ones((128, 2), dtype=np.float32)
    p = engine.Plan(0, 1000, 128, width=128)
    assert p.n_windows == 0 and p.run_host(x.tobytes()).shape == (0, 128)
    # len == W + 1: one window
    x = np.arange(129 * 2, dtype=np.float32).reshape(-1, 2)
    p = engine.Plan(0, 1000, 129, width=128)
    got = p.run_host(x.tobytes())
    ref, _ = oracle.Chain.from_bytes(x.tobytes(), 0, 1000).spark_fft(128, 128)
    assert p.n_windows == 1 and bits_equal(got, ref)
    # a stream shorter than one vector of its format (3 cs8 samples), width 1 and 2
    b = np.array([1, 255, 128, 7, 0, 64], dtype=np.uint8).tobytes()
    for W in (1, 2):
        p = engine.Plan(1, 1000, 3, width=W, stride=1)
        ref, _ = oracle.Chain.from_bytes(b, 1, 1000).spark_fft(W, 1)
        assert bits_equal(p.run_host(b), ref)
    # lowpass whose decimated length equals W + 1
    n = 40 + 8 * 16
    x = (np.random.default_rng(0).standard_normal((n, 2))).astype(np.float32)
    ch = oracle.Chain.from_bytes(x.tobytes(), 0, 1000).lowpass(100, 8, 40)
    assert ch.len() == 17
    p = engine.Plan(0, 1000, n, lowpass=(100, 8, 40), width=16)
    assert p.n_windows == 1 and bits_equal(p.run_host(x.tobytes()), ch.spark_fft(16, 16)[0])


@pytest.mark.parametrize("fmt,lp,W,S,shift", [(0, (1_500_000, 12, 48), 256, 256, -1_250_000),     # D not a power of two
                                              (3, (300_000, 16, 100), 64, 32, 99_000),           # cs16, overlapping windows
                                              (2, None, 32, 32, 5_000),                          # cu8, no lowpass
                                              # FIR-dominated shapes (T >= 8 D): large 512-thread tiles, scalar accumulate chains
                                              (1, (150_000, 64, 800), 32, 8, -40_000),
                                              (0, (1_000_000, 8, 256), 128, 32, 310_000),
                                              (0, (900_000, 16, 256), 512, 512, 77_000)])
def test_plan_time_specialisation_matches_generic(engine, oracle, monkeypatch, fmt, lp, W, S, shift):
    """hiprtc: the same kernel source with the plan's geometry as compile-time constants must give
    the generic kernel's bits (and the oracle's)."""
    from test_gpu_parity import _signal, _to_format, assert_norms_close
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only (no plan-time builds)")
    N = 300_000
    data = _to_format(_signal(np.random.default_rng(W), N), fmt)
    plans = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("QD_JIT", mode)
        plans[mode] = engine.Plan(fmt, 21_000_000, N, shift_hz=shift, lowpass=lp, width=W, stride=S)
    # (chains without a lowpass whose windows lie side by side have a built-in runtime-width kernel of their own: kind 1, not the generic 0)
    assert plans["0"].info.kernel_kind == (1 if lp is None and S == W else 0) and plans["1"].info.kernel_kind == 2
    if lp and lp[2] >= 8 * lp[1]:
        # long-filter policy (512 threads, one large tile) or, for overlapping windows whose geometry allows, the three-stage kernel
        pipe3 = bool(plans["1"].info.kernel_flags & 32768)
        assert plans["1"].info.threads in ((768, 1024) if pipe3 else (512,)) and plans["1"].info.tile_windows >= plans["0"].info.tile_windows
        assert not pipe3 or S < W, plans["1"].info.kernel_flags
    a, b = plans["0"].run_host(data), plans["1"].run_host(data)
    assert bits_equal(a, b)
    ch = oracle.Chain.from_bytes(data, fmt, 21_000_000).shift(shift)
    if lp:
        ch = ch.lowpass(*lp)
    ref, _ = ch.spark_fft(W, S, max_windows=300)
    assert_norms_close(ref, b[:ref.shape[0]], "jit")


@pytest.mark.parametrize("fmt,lp,W,S,shift,hint", [
    # the built-in cfg3' kernel's variant set on a short stream: packed pair FIR + row-aligned phase 1 + deferred FFT (two slots)
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 1, 8, 4, 2, 2 | (76 << 8), 0)),
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 1, 8, 4, 2, 1 | (12 << 8), 0)),
    (3, (200_000, 32, 200), 128, 128, -77_000, (1, 256, 1, 8, 4, 2, 2 | (68 << 8), 0)),     # cs16: deferred FFT without fast phase 1
    (1, (400_000, 16, 64), 64, 64, 120_000, (2, 512, 1, 8, 4, 2, 2 | (68 << 8), 0)),       # two windows per tile, 512 threads
    # cfg4's: two outputs per lane as packed straight-line code, truncated outputs as snapshots; + deferred FFT on the first idle wave
    (0, (5_000_000, 8, 512), 1024, 1024, None, (1, 1024, 2, 4, 4, 2, 1 | (128 << 8), 0)),
    (0, (5_000_000, 8, 512), 1024, 1024, None, (1, 1024, 2, 4, 4, 2, 2 | (192 << 8), 0)),
    (0, (5_000_000, 8, 512), 1024, 1024, None, (1, 1024, 2, 4, 4, 2, 2 | (200 << 8), 0)),     # the built-in cfg4 kernel's set
    (0, (2_000_000, 16, 256), 256, 256, 310_000, (1, 256, 2, 4, 4, 2, 2 | (192 << 8), 0)),  # other D / T / W, shift on, 256 threads
    (2, (1_000_000, 8, 128), 128, 128, -5_000, (2, 512, 2, 4, 4, 2, 1 | (128 << 8), 0)),    # cu8, G = 2
    # round 3: the built-in cfg3' set with non-temporal stream loads (bit 8)
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 1, 8, 4, 2, 2 | (332 << 8), 0)),
    # the 4-aligned packed two-output FIR (T = 200: c = 100 is not 8-aligned), taps of output 1 re-read instead of held (D = 32)
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 2, 8, 4, 2, 2 | (200 << 8), 0)),
    (0, (200_000, 32, 200), 128, 128, None, (1, 256, 2, 8, 4, 2, 1 | (128 << 8), 0)),
    (3, (700_000, 16, 72), 128, 128, 90_000, (1, 256, 2, 8, 4, 2, 2 | (192 << 8), 0)),      # cs16, D = 16 (lag 4: taps held), T/2 = 36
    # the role-split kernel (k_chain_pipe): four producer waves + the FIR wave (320 threads), and + an FFT wave (384 threads)
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 1, 8, 5, 2, 1 | (516 << 8), 0)),
    (0, (200_000, 32, 200), 128, 128, 280_000, (1, 256, 1, 8, 6, 2, 2 | (1540 << 8), 0)),
    (0, (2_000_000, 16, 40), 128, 128, 280_000, (1, 256, 1, 8, 6, 2, 2 | (1796 << 8), 0)),    # cfg2's shape: rows 0-1 / 2 / 3-4, nt loads
    (1, (2_000_000, 16, 40), 128, 128, None, (1, 256, 1, 8, 5, 2, 1 | (516 << 8), 0)),        # cs8 rows of 1024 samples, no shift
    # the built-in cfg2 set (row-aligned phase 1 + nt loads, two windows per tile): 39 windows = a SHORT last tile through the fast path
    (0, (2_000_000, 16, 40), 128, 128, 280_000, (2, 256, 1, 8, 4, 1, 1 | (264 << 8), 0)),
    (0, (2_000_000, 16, 40), 128, 128, 280_000, (2, 256, 1, 8, 4, 1, 1 | (65800 << 8), 0)),   # + shared rows at the default policy (bit 16)
    # half-window tiles (bit 13): two passes per window into one FFT slot, two workgroups per CU — cfg4's shape, and a 256-point
    # window through the packed lane-per-output FIR with a shift stage
    (0, (5_000_000, 8, 512), 1024, 1024, None, (1, 512, 2, 4, 4, 2, 2 | (8392 << 8), 0)),
    (0, (5_000_000, 8, 512), 1024, 1024, None, (1, 512, 2, 4, 4, 2, 2 | (8648 << 8), 0)),
    (0, (200_000, 32, 200), 256, 256, 280_000, (1, 256, 1, 8, 4, 2, 2 | (8268 << 8), 0)),
    # row-aligned phase 1 where only G*S*D (not S*D) is a multiple of the row: cs8, 24 windows of stride 16 per tile (3 rows of 4096)
    (1, (200_000, 32, 400), 64, 16, 280_000, (24, 1024, 1, 8, 4, 2, 1 | (40 << 8), 0)),
    # the three-stage kernel for overlapping windows (k_chain_pipe3, bit 15): producers / shared FIR / FFT waves on consecutive tiles
    (1, (200_000, 32, 400), 64, 16, 280_000, (12, 512, 1, 8, 4, 2, 1 | (32800 << 8), 0)),     # cfg3's chain: cs8, 12 windows per tile
    (0, (200_000, 32, 400), 64, 16, 280_000, (12, 512, 1, 8, 4, 2, 1 | (33056 << 8), 0)),     # the README FSK chain: cf32, nt loads
    (3, (300_000, 16, 128), 64, 32, None, (4, 512, 1, 8, 4, 2, 1 | (32800 << 8), 0)),         # cs16, other taps / stride, no shift (4 windows = one row of 2048)
    # its STREAMING form (k_chain_pipe3s, bits 15 + 17): contiguous runs of tiles per workgroup, samples and outputs carried in LDS rings
    (1, (200_000, 32, 400), 64, 16, 280_000, (12, 512, 1, 8, 4, 2, 1 | (164128 << 8), 0)),    # the built-in cs8 set (39 windows: runs of one tile, a short last one)
    (0, (200_000, 32, 400), 64, 16, 280_000, (14, 512, 1, 8, 4, 2, 1 | (164128 << 8), 0)),    # the built-in cf32 set
    (0, (200_000, 32, 400), 64, 16, 280_000, (6, 512, 1, 8, 4, 2, 1 | (163872 << 8), 0)),     # six-window steps (the smallest the geometry admits here): 96 FIR lanes
    (3, (300_000, 16, 128), 64, 32, None, (4, 512, 1, 8, 4, 2, 1 | (163872 << 8), 0)),        # cs16, no shift
    (1, (200_000, 32, 400), 64, 16, 280_000, (14, 256, 1, 8, 4, 2, 1 | (164128 << 8), 0)),    # the built-in cs8 set: four producer waves (768 threads), 14-window steps
    (0, (200_000, 32, 200), 128, 128, 280_000, (2, 512, 1, 8, 4, 2, 1 | (164100 << 8), 0)),   # windows side by side (S == W): cfg3's chain through the streaming kernel, no trc ring
])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_kernel_variant_flags_equal_generic(engine, oracle, fmt, lp, W, S, shift, hint, epi):
    """Every FixedGeo FLAGS_ variant the built-in kernels use (and neighbours of them on other shapes), forced through
    qd_plan_options.tile_hint, against the generic kernel: same bytes for f32 norms, glyph codes and bucket digits — the
    deferred-FFT path has its own wave-local FFT and epilogues (the bucket one in place), the packed-tile FIR its own
    snapshot logic.  Ragged last tile included (the window count is not a multiple of the tile)."""
    from test_gpu_parity import _signal, _to_format
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only")
    D, T = lp[1], lp[2]
    N = (38 * S + W) * D + T + 3              # 39 windows: a ragged last tile for G = 2
    data = _to_format(_signal(np.random.default_rng(W + T + epi), N), fmt)
    kw = dict(shift_hz=shift, lowpass=lp, width=W, stride=S, epilogue=epi)
    if epi == 1:
        kw["rng"] = (0.001, 0.5)
    ref_plan = engine.Plan(fmt, 21_000_000, N, kernel_policy=engine.KERNEL_GENERIC, **kw)
    var_plan = engine.Plan(fmt, 21_000_000, N, tile_hint=list(hint), **kw)
    assert ref_plan.info.kernel_kind == 0 and var_plan.info.kernel_kind == 2
    flags = hint[6] >> 8
    threads = hint[1] + (512 if flags & 32768 else (0 if not flags & 512 else (128 if flags & 1024 else 64)))       # the role-split kernels add their consumer waves
    assert var_plan.info.tile_windows == hint[0] and var_plan.info.threads == threads
    a, b = ref_plan.run_host(data), var_plan.run_host(data)
    assert a.shape == b.shape and a.shape[0] in (38, 39)         # bucket: lim / stride windows (src/fft.rs:86), one fewer than sparkfft counts here
    assert a.tobytes() == b.tobytes(), (np.nonzero((a != b).reshape(a.shape[0], -1).any(axis=1))[0][:8],)
    # a window sub-range that starts inside a tile (and off the row grid of the row-aligned variants)
    sub = var_plan.run_host(data, 5, 20)
    assert sub.tobytes() == a[5:25].tobytes()
    if epi == 0 and shift is None:
        ref, _ = oracle.Chain.from_bytes(data, fmt, 21_000_000).lowpass(*lp).spark_fft(W, S, max_windows=40)
        assert bits_equal(ref, b[:ref.shape[0]])


def test_random_shapes_specialised_equals_generic(engine, oracle):
    """Fuzz: ~30 random chain shapes (all four formats, odd decimations, overlapping / gapped windows, 2..800 taps,
    some forced onto the register-tiled / wide-workgroup / 16-byte-row variants) — the plan-time specialised kernel
    must reproduce the generic kernel bit for bit (or both must refuse the shape), and the first windows must match the
    CPU oracle (bit-exact without a shift, 4 ulp of the window maximum with one)."""
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only")
    import quadrs_amd as Q
    from util import fuzz_chain_shapes
    checked, bad = fuzz_chain_shapes(Q, 36, 20260101, oracle=oracle)
    assert checked >= 25 and not bad, bad


def test_unhinted_plans_select_the_fast_variants(engine, oracle):
    """the variants of the built-in kernels (packed FIR, row-aligned phase 1, deferred FFT, packed two-output
    tile) are chosen at plan time from the chain's GEOMETRY.  Random shapes from those families with NO tile hint, specialised
    build against the generic kernel bit for bit and against the oracle; most of them must come out with non-zero variant flags."""
    import quadrs_amd as Q
    from util import fuzz_chain_shapes
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only (no variants to select)")
    stats = []
    checked, bad = fuzz_chain_shapes(Q, 28, 20261004, oracle=oracle, auto_only=True, stats=stats)
    assert checked >= 20 and not bad, bad
    flagged = [f for k, f in stats if k == 2 and f != 0]
    assert len(flagged) >= checked // 3, stats       # the rest: tiles beyond half the LDS, windows of 512+ outputs with short filters, 1024-point windows at D = 32
    # a shape next to the north_star one (192 taps) and a long-window one: the expected flag sets
    p = Q.Plan(0, 21_000_000, 1 << 22, shift_hz=280000, lowpass=(200_000, 32, 192), width=128, kernel_policy=Q.KERNEL_SPECIALISE)
    assert p.info.kernel_kind == 2 and p.info.kernel_flags == (4 | 8 | 64 | 256 | 65536) and p.info.threads == 256, (p.info.kernel_kind, p.info.kernel_flags)
    p = Q.Plan(0, 100_000_000, 1 << 24, lowpass=(5_000_000, 8, 384), width=1024, kernel_policy=Q.KERNEL_SPECIALISE)
    assert p.info.kernel_kind == 2 and p.info.kernel_flags == (128 | 64 | 8 | 256 | 8192) and p.info.threads == 512, (p.info.kernel_kind, p.info.kernel_flags)      # half-window tiles
    # overlapping windows with a long filter: the streaming three-stage kernel with the largest step that fits (16 windows of stride 16 = 256 FIR lanes)
    p = Q.Plan(0, 21_000_000, 1 << 24, shift_hz=280000, lowpass=(200_000, 16, 400), width=64, stride=16, kernel_policy=Q.KERNEL_SPECIALISE)
    assert p.info.kernel_kind == 2 and p.info.kernel_flags == (32 | 256 | 32768 | 131072) and p.info.threads == 1024 and p.info.tile_windows == 16, (p.info.kernel_flags, p.info.tile_windows)
    # the policy is not consulted for chains the generic policy must keep: short filters
    p = Q.Plan(0, 21_000_000, 1 << 22, lowpass=(2_000_000, 16, 24), width=128, kernel_policy=Q.KERNEL_SPECIALISE)
    assert p.info.kernel_flags == 0


@pytest.mark.parametrize("lp,W,shift,want_flags", [((200_000, 32, 200), 128, 280_000, 4 | 8 | 64 | 256 | 65536 | 16384),           # the north_star shape
                                                  ((5_000_000, 8, 512), 1024, None, 128 | 64 | 8 | 256 | 8192 | 16384),  # cfg4's shape
                                                  ((2_000_000, 16, 40), 128, 280_000, 8 | 256 | 65536)])                        # a 40-tap filter: nothing to fuse in, the exact built-in kernel
def test_fast_mode_is_opt_in_and_bounded(engine, oracle, lp, W, shift, want_flags):
    """QD_MODE_FAST (SURVEY 8(b) `mode{EXACT_ORDER / FAST}`): a permission to fuse the FIR's multiply-adds.  Never the default; the
    plan says whether its kernel fuses (kernel_flags bit 14); a fused run stays within a few ulp of the window maximum of the exact
    run, and is no further from the infinitely precise filter (f64 FIR + f64 DFT of the same f32 shifted samples) than the exact
    run is."""
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only")
    import quadrs_amd as Q
    from test_gpu_parity import _signal, assert_norms_close, record_observed
    fc, D, T = lp
    N = (23 * W + W) * D + T + 5
    x = _signal(np.random.default_rng(W + T), N).astype(np.float32)
    data = x.tobytes()
    exact = Q.Plan(0, 21_000_000, N, shift_hz=shift, lowpass=lp, width=W)
    fast = Q.Plan(0, 21_000_000, N, shift_hz=shift, lowpass=lp, width=W, mode=Q.MODE_FAST)
    assert exact.info.kernel_flags & 16384 == 0
    assert fast.info.kernel_flags == want_flags, (fast.info.kernel_kind, fast.info.kernel_flags)
    a, b = exact.run_host(data), fast.run_host(data)
    ref, _ = (oracle.Chain.from_bytes(data, 0, 21_000_000).shift(shift) if shift is not None else oracle.Chain.from_bytes(data, 0, 21_000_000)).lowpass(*lp).spark_fft(W, W)
    assert_norms_close(ref, a, "exact")
    if not want_flags & 16384:
        assert a.tobytes() == b.tobytes()
        return
    scale = np.spacing(np.abs(a).max(axis=1, keepdims=True)).astype(np.float64)
    dev = np.abs(a.astype(np.float64) - b.astype(np.float64)) / scale
    assert dev.max() <= 8.0, dev.max()
    assert not np.array_equal(a, b)                              # it IS a different arithmetic
    # truth for a few windows: f64 FIR over the f32 shifted samples (the Shift stage is f32 by definition), f64 DFT, |X|
    sh = x.reshape(-1, 2).copy()
    if shift is not None:
        sh = oracle.shift_apply(sh, 0, oracle.shift_ratio(shift, 21_000_000))
    z = sh[:, 0].astype(np.float64) + 1j * sh[:, 1].astype(np.float64)
    taps = oracle.taps(fc, 21_000_000, T).astype(np.float64)
    c = T - T // 2
    err_e = err_f = 0.0
    for w in (0, 7, 22):
        dec = np.empty(W, dtype=np.complex128)
        for k in range(W):
            jmax = min(T, (W - k) * D + T // 2)
            s0 = (w * W + k) * D + c
            dec[k] = np.dot(z[s0:s0 + jmax], taps[:jmax])
        truth = np.abs(np.fft.fftshift(np.fft.fft(dec)))
        sc = np.spacing(np.float32(truth.max())).astype(np.float64)
        err_e = max(err_e, np.abs(a[w] - truth).max() / sc)
        err_f = max(err_f, np.abs(b[w] - truth).max() / sc)
    record_observed(f"fast mode W={W} T={T}", fast_vs_exact_ulp=float(dev.max()), exact_vs_truth_ulp=float(err_e), fast_vs_truth_ulp=float(err_f))
    assert err_f <= err_e + 1.0, (err_e, err_f)


def test_plan_time_builds_are_cached_on_disk(tmp_path):
    """A plan-time build lands in $QD_JIT_CACHE; a later process whose stream is far too small to justify a compile
    (auto mode) still gets the specialised kernel from the cache, and the same bits."""
    import subprocess, sys, textwrap
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only")
    code = textwrap.dedent("""
        import sys, zlib, numpy as np
        sys.path.insert(0, %r)
        import quadrs_amd as Q
        N = 200_000
        x = (np.random.default_rng(4).standard_normal((N, 2)) * 0.05).astype(np.float32)
        p = Q.Plan(0, 21_000_000, N, shift_hz=123_456, lowpass=(700_000, 12, 56), width=32, stride=24)
        out = p.run_host(x.tobytes())
        print(p.info.kernel_kind, zlib.crc32(out.tobytes()))
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, QD_JIT_CACHE=str(tmp_path))
    env.pop("QD_TUNE", None)
    def run(jit):
        e = dict(env)
        if jit is None:
            e.pop("QD_JIT", None)
        else:
            e["QD_JIT"] = jit
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        kind, crc = r.stdout.split()[-2:]
        return int(kind), crc
    assert run(None) == (0, run("0")[1])                 # nothing cached, small stream: generic kernel
    assert not list(tmp_path.glob("*.co"))
    kind, crc = run("1")                                 # forced build -> cache file
    assert kind == 2 and len(list(tmp_path.glob("*.co"))) == 1
    assert run(None) == (2, crc)                         # auto mode now finds it
    assert run("0")[1] == crc                            # and the generic kernel agrees bit for bit


def test_device_buffers_through_the_abi_only(engine):
    """A host without torch or a HIP toolchain: qd_device_alloc / qd_gen into HBM / qd_plan_run device -> device /
    qd_device_copy back — equal, bit for bit, to the host-buffer path on the same generated stream; plus the argument
    checks of the three device calls."""
    from quadrs_amd import _ffi
    L, check = _ffi.lib(), _ffi.check
    sr, n = 2_000_000, 300_000
    tones = np.array([150_000, -420_000, 731_000], dtype=np.int64)
    host = engine.gen(tones, sr, 0, n)                                   # same kernel, host destination
    p = engine.Plan(0, sr, n, shift_hz=-50_000, lowpass=(200_000, 8, 64), width=64, stride=32)
    want = p.run_host(host.tobytes())
    src, dst = C.c_void_p(), C.c_void_p()
    ob = p.n_windows * 64 * 4
    check(L.qd_device_alloc(n * 8, C.byref(src)))
    check(L.qd_device_alloc(ob, C.byref(dst)))
    try:
        check(L.qd_gen(tones.ctypes.data_as(C.c_void_p), tones.size, sr, 0, n, src, _ffi.MEM_DEVICE))
        check(L.qd_plan_run(p._h, src, _ffi.MEM_DEVICE, 0, n, 0, p.n_windows, dst, _ffi.MEM_DEVICE, None))
        got = np.zeros((p.n_windows, 64), dtype=np.float32)
        check(L.qd_device_copy(got.ctypes.data_as(C.c_void_p), _ffi.MEM_HOST, dst, _ffi.MEM_DEVICE, ob))   # synchronises
        assert bits_equal(got, want)
        back = np.zeros((n, 2), dtype=np.float32)
        check(L.qd_device_copy(back.ctypes.data_as(C.c_void_p), _ffi.MEM_HOST, src, _ffi.MEM_DEVICE, n * 8))
        assert bits_equal(back, host)
        assert L.qd_device_copy(None, _ffi.MEM_HOST, src, _ffi.MEM_DEVICE, 8) == 1            # QD_ERR_INVALID
        assert L.qd_device_copy(back.ctypes.data_as(C.c_void_p), 7, src, _ffi.MEM_DEVICE, 8) == 1
        assert L.qd_device_alloc(16, None) == 1
        z = C.c_void_p(1)
        assert L.qd_device_alloc(0, C.byref(z)) == 0 and not z.value                          # empty buffer: NULL, no error
    finally:
        L.qd_device_free(src); L.qd_device_free(dst)
    assert L.qd_device_free(None) == 0


# ------------------------------------------------------------------ round 2: ABI hardening

def test_two_live_plans_with_different_lds(engine):
    """The generic kernels are process-global: a later plan with a smaller tile must not lower the dynamic-LDS limit under a
    live plan with a larger one (the limit is set to the hardware maximum, not to a plan's tile)."""
    rng = np.random.default_rng(77)
    N = 400_000
    x = (rng.standard_normal((N, 2)) * 0.05).astype(np.float32).tobytes()
    generic = dict(kernel_policy=engine.KERNEL_GENERIC)
    big = engine.Plan(0, 21_000_000, N, shift_hz=99_000, lowpass=(300_000, 16, 400), width=512, stride=512, **generic)
    assert big.info.lds_bytes > 64 * 1024 and big.info.kernel_kind == 0
    want = big.run_host(x)
    small = engine.Plan(0, 21_000_000, N, shift_hz=99_000, lowpass=(300_000, 4, 16), width=16, stride=16, **generic)
    assert small.info.lds_bytes < 48 * 1024
    small.run_host(x)
    engine.fft_norm_batch(np.zeros((64, 2), np.float32), 16, 4, 16)          # a throw-away plan on the same kernels
    assert bits_equal(big.run_host(x), want)
    assert bits_equal(small.run_host(x), small.run_host(x))


@pytest.mark.parametrize("fmt,lp,W,S", [(0, (2_000_000, 16, 40), 128, 128), (1, (200_000, 32, 400), 64, 16), (0, (200_000, 32, 200), 128, 128)])
def test_host_path_many_chunks_equals_device_path(engine, fmt, lp, W, S):
    """The double-buffered host ring with 1 MiB chunks (dozens of chunks, both slots, both streams, the 8-sample slab-start
    rounding, one NCO row table per slot) must give the bytes of a single device-resident launch; pinned host memory
    (QD_MEM_HOST_PINNED, no staging copy) and a mixed pinned-source / pageable-sink run as well."""
    import torch
    from test_gpu_parity import _signal, _to_format
    N = 3_000_000 if fmt == 0 else 9_000_000
    data = np.frombuffer(_to_format(_signal(np.random.default_rng(N + W), N), fmt), dtype=np.uint8)
    dev_plan = engine.Plan(fmt, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W, stride=S)
    src = torch.from_numpy(data.copy()).cuda()
    out = torch.empty(dev_plan.n_windows, W, dtype=torch.float32, device="cuda")
    dev_plan.run_device(src, out)
    torch.cuda.synchronize()
    want = out.cpu().numpy()
    p = engine.Plan(fmt, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W, stride=S, chunk_bytes=1 << 20)
    got = p.run_host(data)
    st = p.stats()
    assert st.chunks >= 5 and st.bytes_h2d >= 0.99 * data.size and st.bytes_d2h == want.nbytes and st.wall_ms > 0 and st.stage_ms > 0
    assert bits_equal(got, want)
    pin_in, pin_out = engine.PinnedBuffer(data.size), engine.PinnedBuffer(want.nbytes)
    pin_in.array[:] = data
    got_p = p.run_host(pin_in.array, pinned=True, out=pin_out.array)
    assert p.stats().stage_ms == 0.0
    assert bits_equal(got_p, want)
    got_m = p.run_host(pin_in.array, pinned=True)                       # pinned source, pageable sink
    assert bits_equal(got_m, want)
    # a sub-range of windows from a slab that starts mid-stream
    w0, nw = dev_plan.n_windows // 3, dev_plan.n_windows // 2
    first, count = p.src_range(w0, nw)
    bps = {0: 8, 1: 2}[fmt]
    assert bits_equal(p.run_host(data[first * bps:(first + count) * bps], w0, nw, src_first=first), want[w0:w0 + nw])
    pin_in.close(); pin_out.close()


@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_sharded_runs_equal_single_device_run(engine, n_shards):
    """Multi-GPU behind the C ABI (qd_plan_run_sharded / _sharded_device): G logical shards, all mapped onto device 0 on a
    one-GPU box (onto distinct devices where there are several), must give the G = 1 bytes — seams included."""
    import torch
    from test_gpu_parity import _signal, _to_format
    n_dev = C.c_int(0)
    engine._ffi.check(engine._ffi.lib().qd_device_count(C.byref(n_dev)))
    devices = [g % n_dev.value for g in range(n_shards)]
    # (the third case is cfg5's geometry — BASELINE configs[4], the 8-GPU workload: cf32, 400 taps / 32, 64-point windows, stride 16 — on
    # the built-in streaming kernel: with n_shards = 8 this is the node's partition mapped onto one device)
    for fmt, lp, W, S, N in ((0, (2_000_000, 16, 40), 128, 128, 2_500_000), (1, (200_000, 32, 400), 64, 16, 4_000_000), (0, (200_000, 32, 400), 64, 16, 6_000_000)):
        data = np.frombuffer(_to_format(_signal(np.random.default_rng(N), N), fmt), dtype=np.uint8)
        one = engine.Plan(fmt, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W, stride=S)
        want = one.run_host(data)
        p = engine.Plan(fmt, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W, stride=S, shard_devices=devices, chunk_bytes=1 << 20)
        infos = [p.shard_info(g) for g in range(n_shards)]
        assert infos[0].w0 == 0 and infos[-1].w1 == p.n_windows and all(a.w1 == b.w0 for a, b in zip(infos, infos[1:]))
        assert all(a.own_first + a.own_count == b.own_first for a, b in zip(infos, infos[1:]) if b.w1 > b.w0)
        halo = (W - S) * lp[1] + lp[2]
        assert all(si.halo == halo for si in infos[:-1] if si.w1 > si.w0) and infos[-1].halo == 0
        assert bits_equal(p.run_sharded_host(data), want)
        # device-resident, pre-split: every shard's slab holds only what it owns (+ room for the halo, fetched peer to peer)
        bps = {0: 8, 1: 2}[fmt]
        slabs, outs = [], []
        for si in infos:
            with torch.cuda.device(si.device):
                buf = torch.zeros((si.own_count + si.halo) * bps + 16, dtype=torch.uint8, device="cuda")
                buf[: si.own_count * bps] = torch.from_numpy(data[si.own_first * bps:(si.own_first + si.own_count) * bps].copy()).cuda()
                slabs.append(buf)
                outs.append(torch.empty(max(si.w1 - si.w0, 1), W, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        p.run_sharded_device([t.data_ptr() for t in slabs], [t.data_ptr() for t in outs], sync=True)
        got = np.concatenate([o.cpu().numpy()[: si.w1 - si.w0] for o, si in zip(outs, infos)])
        assert bits_equal(got, want)


@pytest.mark.parametrize("n_shards", [3, 8])
def test_sharded_lowpass_free_plans(engine, n_shards):
    """The wave-local kernel family behind qd_plan_run_sharded: shards of a lowpass-free chain (windows side by side, overlapping windows as
    interleaved launches with and without a shift, a window per lane) start on the windows the plan's tile_windows names, so every shard
    runs the same kernels on the same NCO row grids as the one-device run: the same bytes, seams included."""
    from quadrs_amd import _ffi
    from test_gpu_parity import _signal, _to_format
    n_dev = C.c_int(0)
    engine._ffi.check(engine._ffi.lib().qd_device_count(C.byref(n_dev)))
    devices = [g % n_dev.value for g in range(n_shards)]
    for fmt, shift, W, S, N in ((0, 280000, 256, 256, 700_000), (0, 280000, 64, 16, 500_000), (1, None, 4, 2, 300_000), (3, None, 32, 8, 400_000), (0, None, 512, 128, 900_000)):
        data = np.frombuffer(_to_format(_signal(np.random.default_rng(N), N), fmt), dtype=np.uint8)
        kw = dict(shift_hz=shift, width=W, stride=S, kernel_policy=_ffi.KERNEL_SPECIALISE)
        one = engine.Plan(fmt, 21_000_000, N, **kw)
        assert one.info.kernel_flags & 524288
        want = one.run_host(data)
        p = engine.Plan(fmt, 21_000_000, N, shard_devices=devices, chunk_bytes=1 << 20, **kw)
        infos = [p.shard_info(g) for g in range(n_shards)]
        assert infos[0].w0 == 0 and infos[-1].w1 == p.n_windows and all(a.w1 == b.w0 for a, b in zip(infos, infos[1:]))
        assert all(si.w0 % one.info.tile_windows == 0 for si in infos)
        assert all(si.halo == W - S for si in infos[:-1] if si.w1 > si.w0)
        assert bits_equal(p.run_sharded_host(data), want), (fmt, shift, W, S)


def test_fine_grained_calls_reuse_workspaces(engine, oracle):
    """One fine-grained call per window, as the read_at shims of INTEGRATION.md make them: no allocation per call (device
    memory in use stays flat once the pool is warm), results unchanged, and qd_set_stream moves the work to a caller's stream."""
    import torch
    L = engine._ffi.lib()
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((4096 + 40, 2)) * 0.1).astype(np.float32)
    taps = engine.lowpass_design(2_000_000, 21_000_000, 40)
    ratio = engine.shift_ratio(280000, 21_000_000)
    def one_window(off):
        s = engine.shift(x, off, ratio)
        n, dec = engine.lowpass_block(taps, 16, s)
        return engine.fft_norm_batch(dec[:128], 128, 1, 128)
    first = one_window(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(200):
        again = one_window(0 if i % 2 == 0 else 7 * i)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)        # flat: no per-call allocations accumulating
    assert bits_equal(one_window(0), first)
    ref = oracle.norm(oracle.fft(oracle.lowpass_block(oracle.taps(2_000_000, 21_000_000, 40), 16, oracle.shift_apply(x, 0, ratio))[1][:128]))[np.r_[64:128, 0:64]]
    assert (np.abs(first[0].astype(np.float64) - ref) <= np.spacing(ref.max())).all()
    st = torch.cuda.Stream()
    engine._ffi.check(L.qd_set_stream(C.c_void_p(st.cuda_stream)))
    try:
        assert bits_equal(one_window(0), first)
        d = torch.from_numpy(x.copy()).cuda()
        torch.cuda.synchronize()
        engine._ffi.check(L.qd_shift(C.c_void_p(d.data_ptr()), d.shape[0], 0, ratio, engine.MEM_DEVICE))   # async on st
        st.synchronize()
        assert bits_equal(d.cpu().numpy(), engine.shift(x, 0, ratio))
    finally:
        engine._ffi.check(L.qd_set_stream(None))
    engine._ffi.check(L.qd_release_workspaces())
    assert bits_equal(one_window(0), first)                          # the pool refills itself


@pytest.mark.parametrize("W,out_len", [(5, 64), (12, 64), (100, 48), (127, 32), (1000, 16), (1536, 8), (4095, 3), (3000, 4), (4093, 2)])
def test_take_fft_any_width_against_f64_dft(engine, oracle, fsk, W, out_len):
    """A8 / N3: take_fft at widths that are not powers of two (the reference's planner takes any, src/ffts.rs:25; slider
    4..4096, src/eui/mod.rs:157).  rustfft's result for these lengths depends on its planner / host-SIMD code path => PARITY
    UNPINNED; correctness here is the mathematical one.  The kernel carries the Bluestein convolution in f64, so every bin is
    the exact DFT of the (f32-windowed) samples rounded to Complex<f32>, then hypotf — the oracle does the same from an f64 DFT.
    Bound: 2 ulp_f32 of the reference norm plus 1e-12 of the row's l1 norm (bins that cancel to ~0) — no log2(W) factor."""
    from test_gpu_parity import record_observed
    x = np.frombuffer(fsk, dtype=np.float32).reshape(-1, 2)
    for windowing in (0, 1):
        rc, ref, offs = oracle.Chain.from_bytes(fsk, oracle.FMT_CF32, 21_000_000).take_fft(W, out_len, None, windowing)
        assert rc == 0
        got = engine.take_fft(x, W, out_len, None, windowing)
        assert got.shape == ref.shape
        worst_ulp, exact = 0.0, 0
        for r in range(out_len):
            seg = x[int(offs[r]):int(offs[r]) + W].astype(np.float64)
            l1 = float(np.abs(seg).sum())
            err = np.abs(got[r].astype(np.float64) - ref[r].astype(np.float64))
            allowed = 2.0 * np.spacing(ref[r]).astype(np.float64) + 1e-12 * l1
            assert (err <= allowed).all(), (W, windowing, r, float((err / allowed).max()))
            worst_ulp = max(worst_ulp, float((err / np.spacing(ref[r]).astype(np.float64)).max()))
            exact += int((got[r].view(np.uint32) == ref[r].view(np.uint32)).sum())
        record_observed(f"take_fft W={W} windowing={windowing}", rows=out_len, worst_ulp=worst_ulp, exact_fraction=exact / (out_len * W))


def test_streaming_three_stage_kernel_long_runs(engine):
    """k_chain_pipe3s at a size where every workgroup owns a run of ~43 steps (10 923 tiles on 256 workgroups: uneven runs, a short last
    tile), device path, with a window sub-range that starts on a row boundary inside the stream: bit for bit against the generic
    kernel, twice (the rings wrap ~7 times per run).  cs8 built-in and a cf32 plan-time build."""
    if os.environ.get("QD_NO_FIXED"):
        pytest.skip("QD_NO_FIXED=1 runs the generic kernels only")
    import torch
    import bench
    dev = torch.device("cuda", 0)
    for fmt, hint in ((1, None), (0, [14, 512, 1, 8, 4, 2, 1 | (164128 << 8), 0])):
        n = (1 << 26) - 40_000                                # a window count that is no multiple of the step (14)
        src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
        kw = dict(shift_hz=280000, lowpass=(200_000, 32, 400), width=64, stride=16)
        ref = engine.Plan(fmt, 21_000_000, n, kernel_policy=engine.KERNEL_GENERIC, **kw)
        var = engine.Plan(fmt, 21_000_000, n, **(dict(tile_hint=hint) if hint else {}), **kw)
        assert var.info.kernel_flags == 164128 and var.info.threads == (768 if fmt == 1 else 1024) and ref.n_windows % var.info.tile_windows != 0      # cs8: four producer waves
        a = torch.empty(ref.n_windows, 64, dtype=torch.float32, device=dev)
        b = torch.zeros_like(a)
        ref.run_device(src, a)
        for _ in range(2):
            b.zero_()
            var.run_device(src, b)
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), fmt
        # a sub-range whose first window sits on the row grid (window 4096 * k: 4096 * 512 samples) and whose length is odd
        w0, nw = 8192, 100_001
        b.zero_()
        var.run_device(src, b[w0:w0 + nw], first_window=w0, n_windows=nw)
        torch.cuda.synchronize()
        assert torch.equal(a[w0:w0 + nw].view(torch.int32), b[w0:w0 + nw].view(torch.int32)), fmt
        assert not b[:w0].any() and not b[w0 + nw:].any()
        var.close(); ref.close()
        del src, a, b


def test_three_stage_kernel_with_the_tile_queue(engine):
    """k_chain_pipe3 (producers / shared FIR / FFT waves on consecutive tiles) at a size where the dynamic tile queue is active
    (10 922 tiles on 256 workgroups) and the last tile is short: bit for bit against the generic kernel, run twice."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    n = 1 << 26
    src = bench.synth_slab(torch, 1, 0, n, 0x5EED0002, dev)
    kw = dict(shift_hz=280000, lowpass=(200_000, 32, 400), width=64, stride=16)
    ref = engine.Plan(1, 21_000_000, n, kernel_policy=engine.KERNEL_GENERIC, **kw)
    var = engine.Plan(1, 21_000_000, n, tile_hint=[12, 512, 1, 8, 4, 2, 1 | (32800 << 8), 0], **kw)
    assert var.info.kernel_flags == 32800 and var.info.threads == 1024 and ref.n_windows % 12 != 0
    a = torch.empty(ref.n_windows, 64, dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    ref.run_device(src, a)
    for _ in range(2):
        b.zero_()
        var.run_device(src, b)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


# ------------------------------------------------------------------ chains without a lowpass: the wave-local kernel (k_spark)

def _synth_bytes(fmt, n, seed):
    rng = np.random.default_rng(seed)
    if fmt == 0:
        t = np.arange(n)
        z = 0.3 * np.exp(2j * np.pi * (0.07 * t + 0.00002 * t * t)) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        return np.stack([z.real, z.imag], 1).astype(np.float32).tobytes()
    if fmt == 3:
        return rng.integers(-32768, 32768, size=(n, 2), dtype=np.int64).astype(np.int16).tobytes()
    return rng.integers(0, 256, size=(n, 2), dtype=np.int64).astype(np.uint8).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,shift", [(0, None), (0, 280000), (1, None), (1, 280000), (2, -1_234_567), (3, None), (3, 280000)])
def test_wave_local_kernel_equals_generic_and_oracle(engine, oracle, fmt, shift):
    """`from F [shift] sparkfft` with stride == width (README example 1's chain, src/fft.rs:28-65 straight over Shift / SampleFile) runs
    on the wave-local kernel (kernel_flags bit 19).  Every width 1 ... 1024, all three epilogues, a stream that ends inside a tile
    and inside a load vector, window sub-ranges that start on and off the NCO row grid, a slab that starts inside the stream: bit
    for bit the generic kernel's output (QD_KERNEL_GENERIC), and the oracle's (bit-exact without a shift; the NCO's tolerance with)."""
    import torch
    from quadrs_amd import _ffi
    bps = {0: 8, 1: 2, 2: 2, 3: 4}[fmt]
    sr = 21_000_000
    for W in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
        n = 5 * 1024 + 3 * W + (7 if W > 4 else 1)
        data = _synth_bytes(fmt, n, 1000 * fmt + W)
        ch = oracle.Chain.from_bytes(data, fmt, sr)
        if shift is not None:
            ch = ch.shift(shift)
        for epi, rng_ in ((engine.EPI_NORMS_F32, None), (engine.EPI_GLYPH_U8, (0.01, 0.5) if fmt == 0 else (0.3, 30.0)), (engine.EPI_BUCKET2_U8, None)):
            if epi == engine.EPI_BUCKET2_U8 and W < 2:
                continue
            kw = dict(shift_hz=shift, width=W, stride=W, epilogue=epi, rng=rng_)
            p = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_NO_PLAN_TIME, **kw)      # the built-in runtime-width kernel
            g = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_GENERIC, **kw)
            assert p.info.kernel_flags & 524288 and not (g.info.kernel_flags & 524288), (W, p.info.kernel_flags, g.info.kernel_flags)
            assert p.info.kernel_kind == 1
            assert p.n_windows == g.n_windows
            a, b = p.run_host(data), g.run_host(data)
            assert np.array_equal(a, b), (fmt, shift, W, epi, int((a != b).sum()))
            if fmt == 0 or W in (4, 64, 256) or (W in (8, 16, 32, 128, 512, 1024) and epi == engine.EPI_NORMS_F32):
                # the plan-time builds (what a stream of 1 GiB and more gets): width as a compile-time constant, and at W = 128 ... 1024
                # the kernel whose base butterflies run out of the row registers (bit 20; every format)
                j = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_SPECIALISE, **kw)
                assert j.info.kernel_kind == 2 and j.info.kernel_flags & 524288
                assert bool(j.info.kernel_flags & 1048576) == (W in (128, 256, 512, 1024)), (W, j.info.kernel_flags)
                c = j.run_host(data)
                assert np.array_equal(c, b), (fmt, shift, W, epi, "plan-time build", int((c != b).sum()))
                if epi == engine.EPI_NORMS_F32:
                    nwj = j.n_windows
                    for w0 in sorted({0, min(nwj - 1, max(1, 512 // W)), min(nwj - 1, max(1, 2048 // W) + 1), nwj // 2}):
                        first, count = j.src_range(w0, nwj - w0)
                        sub = j.run_host(data[first * bps:(first + count) * bps], w0, nwj - w0, src_first=first)
                        assert np.array_equal(sub, b[w0:]), (fmt, shift, W, w0, "plan-time build")
                j.close()
            if epi == engine.EPI_NORMS_F32:
                ref, _ = ch.spark_fft(W, W)
                assert ref.shape == a.shape
                if shift is None:
                    assert bits_equal(ref, a), (fmt, W)
                else:
                    mx = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-30)
                    assert (np.abs(a.astype(np.float64) - ref) / np.spacing(mx.astype(np.float32))).max() <= 1.0, (fmt, W)
                    assert (a.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.999
                # window sub-ranges: on the row grid (512 samples), off it, and a device slab that starts inside the stream
                nw = p.n_windows
                for w0 in sorted({0, min(nw - 1, max(1, 512 // W)), min(nw - 1, max(1, 1024 // W) + 1), nw // 2}):
                    cnt = nw - w0
                    first, count = p.src_range(w0, cnt)
                    sub = p.run_host(data[first * bps:(first + count) * bps], w0, cnt, src_first=first)
                    assert np.array_equal(sub, a[w0:]), (fmt, shift, W, w0)
                    src = torch.frombuffer(bytearray(data[first * bps:(first + count) * bps]), dtype=torch.uint8).cuda()
                    out = torch.empty(cnt, W, dtype=torch.float32, device="cuda")
                    p.run_device(src, out, w0, cnt, src_first=first, src_count=count)
                    torch.cuda.synchronize()
                    assert np.array_equal(out.cpu().numpy(), a[w0:]), (fmt, shift, W, w0, "device")
            p.close(); g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [0, 1, 3])
def test_overlapping_windows_without_lowpass_as_interleaved_launches(engine, oracle, fmt):
    """`from F sparkfft -width W -stride S` with S | W, S < W and nothing in front (README example 1: `-width 4 -stride 2`; src/fft.rs:28-65
    over SampleFile): the windows phi, phi + W/S, ... lie side by side in the stream shifted by phi S samples, so a plan-time build of the
    wave-local kernel runs W/S times, each launch writing every (W/S)-th output row.  Against the generic kernel and the oracle bit for
    bit: norms and glyphs, streams that end anywhere, window sub-ranges that start in any phase, slabs that start inside the stream and
    device slabs off the load-vector grid (those take the per-sample kernel)."""
    import torch
    from quadrs_amd import _ffi
    bps = {0: 8, 1: 2, 3: 4}[fmt]
    spl = {0: 2, 1: 4, 3: 4}[fmt]
    sr = 21_000_000
    for W, S in ((2, 1), (4, 2), (8, 4), (8, 6), (16, 4), (16, 2), (32, 8), (64, 16), (64, 24), (128, 64), (128, 48), (256, 32), (256, 200), (512, 256), (1024, 512), (1024, 32)):
        n = 9 * 1024 + 3 * W + 5
        data = _synth_bytes(fmt, n, 77 * fmt + W + S)
        ch = oracle.Chain.from_bytes(data, fmt, sr)
        ref, _ = ch.spark_fft(W, S)
        for epi, rng_ in ((engine.EPI_NORMS_F32, None), (engine.EPI_GLYPH_U8, (0.01, 0.5) if fmt == 0 else (0.3, 30.0)), (engine.EPI_BUCKET2_U8, None)):
            if epi == engine.EPI_BUCKET2_U8 and W not in (4, 16, 64, 128, 1024):
                continue
            kw = dict(width=W, stride=S, epilogue=epi, rng=rng_)
            j = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_SPECIALISE, **kw)
            g = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_GENERIC, **kw)
            # W <= 8: a window per lane (k_spark0), W >= 128: one launch of k_spark2, both at any stride and sink; between:
            # interleaved launches of k_spark when the stride divides the width (norms / glyphs)
            expect_phases = (S * bps) % 4 == 0 and (W <= 8 or W >= 128 or (W >= spl and W % S == 0 and epi != engine.EPI_BUCKET2_U8))
            assert bool(j.info.kernel_flags & 524288) == expect_phases, (fmt, W, S, j.info.kernel_flags)
            if expect_phases:
                assert j.info.kernel_kind == 2
                # W = 128 ... 1024: ONE launch of the register-first-pass kernel built for the stride (the overlap comes out of the caches);
                # below: W / S interleaved launches, any window range (ranges on multiples of tile_windows keep the fast path)
                assert bool(j.info.kernel_flags & 1048576) == (W in (128, 256, 512, 1024))
                assert bool(j.info.kernel_flags & 2097152) == (W <= 8)
                assert j.info.tile_windows == (64 if W <= 8 else max(1, spl // S) if W < 128 else {128: 8, 256: 8, 512: 2, 1024: 2}[W])
            else:
                assert not (j.info.kernel_flags & 524288)
            a, b = j.run_host(data), g.run_host(data)
            assert a.shape == b.shape and np.array_equal(a, b), (fmt, W, S, epi, int((a != b).sum()))
            if epi != engine.EPI_NORMS_F32:
                j.close(); g.close()
                continue
            assert ref.shape == a.shape and bits_equal(ref, a), (fmt, W, S)
            nw, R = j.n_windows, W // S
            for w0 in sorted({1, R - 1, R, R + 1, nw // 2, nw - 1, nw - R - 1}):
                for cnt in sorted({1, min(R + 1, nw - w0), nw - w0}):
                    first, count = j.src_range(w0, cnt)
                    sub = j.run_host(data[first * bps:(first + count) * bps], w0, cnt, src_first=first)
                    assert np.array_equal(sub, a[w0:w0 + cnt]), (fmt, W, S, w0, cnt)
            # a device slab whose first sample is not on the load-vector grid
            w0 = max(1, nw // 3)
            first, count = j.src_range(w0, nw - w0)
            if first % spl == 0:
                first -= 1; count += 1
            src = torch.frombuffer(bytearray(data[first * bps:(first + count) * bps]), dtype=torch.uint8).cuda()
            out = torch.empty(nw - w0, W, dtype=torch.float32, device="cuda")
            j.run_device(src, out, w0, nw - w0, src_first=first, src_count=count)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), a[w0:]), (fmt, W, S, "unaligned slab")
            j.close(); g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [0, 1, 3])
def test_overlapping_windows_with_a_shift_as_interleaved_launches(engine, oracle, fmt):
    """`from F shift X sparkfft -width W -stride S`, S | W, no lowpass: W / S interleaved launches of the wave-local kernels (k_spark, or
    k_spark2's S = W build from 128 points on), launch phi with an NCO row table of its own — rows of 512 samples that start phi S samples
    into the stream (src/shift.rs:46-54 is a function of the absolute sample index).  Another tiling of the same NCO scheme than the
    generic kernel's: compared within the NCO's tolerance with it and with the oracle; window sub-ranges on the launches'
    row grids reproduce the whole run bit for bit, others take the generic kernel."""
    from quadrs_amd import _ffi
    bps = {0: 8, 1: 2, 3: 4}[fmt]
    sr = 21_000_000
    for W, S, shift in ((64, 16, 280000), (16, 4, -1_234_567), (8, 4, 280000), (128, 64, 280000), (256, 32, 99_000), (1024, 256, 280000), (32, 32 // 2, 5_000_000)):
        n = 24 * 1024 + 3 * W + 5
        data = _synth_bytes(fmt, n, 91 * fmt + W + S)
        ch = oracle.Chain.from_bytes(data, fmt, sr).shift(shift)
        ref, _ = ch.spark_fft(W, S)
        for epi, rng_ in ((engine.EPI_NORMS_F32, None), (engine.EPI_GLYPH_U8, (0.01, 0.5) if fmt == 0 else (0.3, 30.0))):
            kw = dict(shift_hz=shift, width=W, stride=S, epilogue=epi, rng=rng_)
            j = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_SPECIALISE, **kw)
            g = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_GENERIC, **kw)
            R = W // S
            eligible = (S * bps) % 4 == 0 and W >= 4
            assert bool(j.info.kernel_flags & 524288) == eligible, (fmt, W, S, j.info.kernel_flags)
            if eligible:
                assert j.info.kernel_kind == 2 and bool(j.info.kernel_flags & 1048576) == (W >= 128)
                assert j.info.tile_windows == R * max(1, 512 // W)
            a, b = j.run_host(data), g.run_host(data)
            assert a.shape == b.shape
            if epi != engine.EPI_NORMS_F32:
                assert (a != b).mean() <= 2e-3, (fmt, W, S, float((a != b).mean()))      # glyph cells next to a threshold
                j.close(); g.close()
                continue
            for other, what in ((b, "generic"), (ref, "oracle")):
                assert other.shape == a.shape
                mx = np.maximum(np.abs(other).max(axis=1, keepdims=True), 1e-30)
                assert (np.abs(a.astype(np.float64) - other) / np.spacing(mx.astype(np.float32))).max() <= 1.0, (fmt, W, S, what)
                assert (a.view(np.uint32) == other.astype(np.float32).view(np.uint32)).mean() >= 0.999, (fmt, W, S, what)
            # the host path in many chunks: chunk boundaries fall on the launches' row grids, so the bytes are the one-chunk run's
            jc = engine.Plan(fmt, sr, n, kernel_policy=_ffi.KERNEL_SPECIALISE, chunk_bytes=1 << 16, **kw)
            assert np.array_equal(jc.run_host(data), a), (fmt, W, S, "chunked host path")
            jc.close()
            nw, unit = j.n_windows, j.info.tile_windows
            for w0 in sorted({unit, 2 * unit, (nw // (2 * unit)) * unit}):
                if w0 >= nw:
                    continue
                for cnt in sorted({1, min(R + 1, nw - w0), nw - w0}):
                    first, count = j.src_range(w0, cnt)
                    sub = j.run_host(data[first * bps:(first + count) * bps], w0, cnt, src_first=first)
                    assert np.array_equal(sub, a[w0:w0 + cnt]), (fmt, W, S, w0, cnt)
            w0 = unit + 1                                                       # off the row grids: the generic kernel, within the tolerance
            if w0 < nw:
                first, count = j.src_range(w0, nw - w0)
                sub = j.run_host(data[first * bps:(first + count) * bps], w0, nw - w0, src_first=first)
                mx = np.maximum(np.abs(a[w0:]).max(axis=1, keepdims=True), 1e-30)
                assert (np.abs(sub.astype(np.float64) - a[w0:]) / np.spacing(mx)).max() <= 1.0, (fmt, W, S, "off-grid range")
            j.close(); g.close()


@pytest.mark.gpu
def test_random_lowpass_free_shapes(engine, oracle):
    """Random `from F [shift] sparkfft -width W -stride S` chains (every format, W = 1 ... 1024, strides up to 2 W, three sinks, whole
    streams and random window sub-ranges): the plan-time builds of the wave-local family against the generic chain kernel and the oracle."""
    import quadrs_amd as Q
    from util import fuzz_nofir_shapes
    stats = []
    checked, bad = fuzz_nofir_shapes(Q, 40, 20261005, oracle=oracle, stats=stats)
    assert checked >= 30 and not bad, bad
    assert sum(1 for k, f in stats if f & 524288) >= 10      # the family under test was actually chosen


@pytest.mark.gpu
def test_wave_local_kernel_many_tiles(engine, oracle):
    """More tiles than resident waves (grid-stride walk, prefetch of the next tile, the last wave's short run): 2^24 cf32 samples,
    W = 128 with and without a shift, against the generic kernel bit for bit and against the oracle on sampled windows."""
    import torch
    from quadrs_amd import _ffi
    import bench
    n = (1 << 24) + 5 * 128 + 3
    dev = torch.device("cuda", 0)
    slab = bench.synth_slab(torch, 0, 0, n, 0x5EED0002, dev)
    host = slab.cpu().numpy().tobytes()
    for shift in (None, 280000):
        p = engine.Plan(0, 21_000_000, n, shift_hz=shift, width=128, stride=128, kernel_policy=_ffi.KERNEL_NO_PLAN_TIME)
        g = engine.Plan(0, 21_000_000, n, shift_hz=shift, width=128, stride=128, kernel_policy=_ffi.KERNEL_GENERIC)
        j = engine.Plan(0, 21_000_000, n, shift_hz=shift, width=128, stride=128, kernel_policy=_ffi.KERNEL_SPECIALISE)
        assert p.info.kernel_flags & 524288 and j.info.kernel_flags & 1048576
        a = torch.empty(p.n_windows, 128, dtype=torch.float32, device=dev)
        b, c = torch.empty_like(a), torch.empty_like(a)
        p.run_device(slab, a); g.run_device(slab, b); j.run_device(slab, c)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert torch.equal(c.view(torch.int32), b.view(torch.int32))
        j.close()
        ch = oracle.Chain.from_bytes(host, 0, 21_000_000)
        if shift is not None:
            ch = ch.shift(shift)
        for w0 in (0, p.n_windows // 3, p.n_windows - 300):
            ref, _ = ch.spark_fft(128, 128, first_window=w0, max_windows=300)
            got = a[w0:w0 + 300].cpu().numpy()
            if shift is None:
                assert bits_equal(ref, got)
            else:
                assert (got.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.9999
        p.close(); g.close()
