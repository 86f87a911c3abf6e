"""ctypes loader for the CPU oracle (oracle/libquadrs_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under quadrs_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("QD_ORACLE_SO") or os.path.join(_HERE, "libquadrs_oracle.so")     # QD_ORACLE_SO: an instrumented build (scripts/sanitize_cpu.sh)

FMT_CF32, FMT_CS8, FMT_CU8, FMT_CS16 = 0, 1, 2, 3
PANIC = C.c_size_t(-1).value
U64_MAX = (1 << 64) - 1

c32 = np.dtype([("re", "<f4"), ("im", "<f4")])


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("quadrs_oracle.c", "quadrs_oracle.h")]
    if os.environ.get("QD_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libquadrs_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, u64, i64, sz, f32, f64, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_size_t, C.c_float, C.c_double, C.c_int
        sig = {
            "qo_pair_bytes": (u64, [i32]),
            "qo_unpack": (None, [i32, vp, sz, vp]),
            "qo_source_mem": (vp, [vp, u64, i32, u64]),
            "qo_source_gen": (vp, [vp, sz, u64, f64]),
            "qo_shift": (vp, [vp, i64]),
            "qo_lowpass": (vp, [vp, u64, u64, sz]),
            "qo_free": (None, [vp]),
            "qo_len": (u64, [vp]),
            "qo_sample_rate": (u64, [vp]),
            "qo_read_at": (sz, [vp, u64, vp, sz]),
            "qo_read_exact_at": (i32, [vp, u64, vp, sz]),
            "qo_set_lowpass_closed_form": (None, [i32]),
            "qo_shift_ratio": (f64, [i64, u64]),
            "qo_shift_multiplier": (None, [f64, u64, vp, vp]),
            "qo_shift_apply": (None, [vp, sz, u64, f64]),
            "qo_cutoff": (f32, [u64, u64]),
            "qo_lowpass_taps": (None, [f32, sz, vp]),
            "qo_complex_convolve": (sz, [vp, sz, vp, sz, vp]),
            "qo_lowpass_block": (sz, [vp, sz, u64, vp, sz, vp, sz]),
            "qo_fft_new": (vp, [sz]),
            "qo_fft_free": (None, [vp]),
            "qo_fft_process": (None, [vp, vp]),
            "qo_fft_twiddle_count": (sz, [vp]),
            "qo_fft_twiddles": (vp, [vp]),
            "qo_fft_base_len": (sz, [vp]),
            "qo_dft_f64": (None, [vp, sz, vp, vp]),
            "qo_spark_window_count": (u64, [u64, u64, u64]),
            "qo_spark_fft": (u64, [vp, sz, u64, i32, f32, f32, u64, u64, vp, vp]),
            "qo_glyph_code": (C.c_uint8, [f32, f32, f32]),
            "qo_spark_render": (sz, [u64, vp, u64, sz, vp, sz]),
            "qo_freq_levels": (u64, [vp, sz, u64, u64, vp]),
            "qo_blackman_harris": (None, [sz, vp]),
            "qo_take_fft": (i32, [vp, i32, u64, u64, sz, i32, sz, vp, vp]),
            "qo_do_write": (i32, [vp, vp, u64, vp]),
            "qo_norm_batch": (None, [vp, sz, vp]),
            "qo_device_norm_model": (f32, [f32, f32, i32, vp]),
            "qo_device_norm_selftest": (u64, [u64, u64, i32, i32, i32, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def as_c32(a):
    """view a complex64 / (n,2) float32 array as contiguous float32 pairs"""
    a = np.ascontiguousarray(a)
    if a.dtype == np.complex64:
        return a.view(np.float32).reshape(-1, 2)
    return a.astype(np.float32, copy=False).reshape(-1, 2)


class Chain:
    """A Samples chain (src/samples.rs:11-28): from/gen -> shift -> lowpass, owned by Python."""

    def __init__(self):
        self._node = None
        self._keep = []

    @classmethod
    def from_bytes(cls, data, fmt, sample_rate):
        self = cls()
        buf = np.frombuffer(bytes(data) if not isinstance(data, np.ndarray) else data.tobytes(), dtype=np.uint8)
        buf = np.ascontiguousarray(buf)
        self._keep.append(buf)
        self._node = lib().qo_source_mem(_p(buf), buf.size, fmt, sample_rate)
        return self

    @classmethod
    def gen(cls, cos_hz, sample_rate, seconds=1.0):
        self = cls()
        arr = np.asarray(cos_hz, dtype=np.int64)
        node = lib().qo_source_gen(_p(arr), arr.size, sample_rate, float(seconds))
        if not node:
            raise ValueError("Gen::new would return Err")
        self._node = node
        return self

    def shift(self, frequency):
        node = lib().qo_shift(self._node, int(frequency))
        if not node:
            raise AssertionError("Shift::new would panic")
        self._node = node
        return self

    def lowpass(self, frequency, decimate=8, size=40):
        self._node = lib().qo_lowpass(self._node, int(frequency), int(decimate), int(size))
        return self

    def __del__(self):
        if self._node and _lib is not None:
            _lib.qo_free(self._node)
            self._node = None

    def len(self):
        return lib().qo_len(self._node)

    def sample_rate(self):
        return lib().qo_sample_rate(self._node)

    def read_at(self, off, n):
        """returns (count, float32[n,2]); count == PANIC where the reference panics"""
        out = np.zeros((n, 2), dtype=np.float32)
        got = lib().qo_read_at(self._node, int(off), _p(out), n)
        return got, out

    def spark_fft(self, width=128, stride=None, rng=None, first_window=0, max_windows=None,
                  want_norms=True, want_codes=True):
        stride = width if stride is None else stride
        total = lib().qo_spark_window_count(self.len(), width, stride)
        if total == U64_MAX:
            raise RuntimeError("spark_fft: len < width (u64 underflow in the reference)")
        cap = max(0, total - first_window)
        if max_windows is not None:
            cap = min(cap, max_windows)
        norms = np.zeros((cap, width), dtype=np.float32) if want_norms else None
        codes = np.zeros((cap, width), dtype=np.uint8) if want_codes else None
        mn, mx = rng if rng else (0.0, 0.0)
        got = lib().qo_spark_fft(self._node, width, stride, 1 if rng else 0, mn, mx, first_window, cap,
                                 _p(norms) if want_norms else None, _p(codes) if want_codes else None)
        if got == U64_MAX:
            raise RuntimeError("spark_fft: read_exact_at failed")
        assert got == cap
        return norms, codes

    def spark_text(self, width=128, stride=None, rng=None):
        _, codes = self.spark_fft(width, stride, rng, want_norms=False)
        return render(self.sample_rate(), codes)

    def freq_levels(self, width=128, stride=None):
        stride = width if stride is None else stride
        ln = self.len()
        cap = (ln - width) // stride
        vals = np.zeros(max(cap, 1), dtype=np.uint8)
        got = lib().qo_freq_levels(self._node, width, stride, cap, _p(vals))
        if got == U64_MAX:
            raise RuntimeError("freq_levels failed")
        return vals[:got]

    def take_fft(self, width, output_len, slice_=None, windowing=1):
        rows = np.zeros((output_len, width), dtype=np.float32)
        offs = np.zeros(output_len, dtype=np.uint64)
        s, e = slice_ if slice_ else (0, 0)
        rc = lib().qo_take_fft(self._node, 1 if slice_ else 0, s, e, width, windowing, output_len, _p(rows), _p(offs))
        return rc, rows, offs

    def do_write(self, cap):
        out = np.zeros((cap, 2), dtype=np.float32)
        n = C.c_uint64(0)
        rc = lib().qo_do_write(self._node, _p(out), cap, C.byref(n))
        return rc, n.value, out[: min(n.value, cap)]


def render(sample_rate, codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    nwin, W = codes.shape
    need = lib().qo_spark_render(sample_rate, _p(codes), nwin, W, None, 0)
    buf = C.create_string_buffer(need)
    lib().qo_spark_render(sample_rate, _p(codes), nwin, W, buf, need)
    return buf.raw[:need]


def unpack(fmt, data):
    data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    n = data.size // lib().qo_pair_bytes(fmt)
    out = np.zeros((n, 2), dtype=np.float32)
    lib().qo_unpack(fmt, _p(data), n, _p(out))
    return out


def taps(frequency, sample_rate, size):
    out = np.zeros(size, dtype=np.float32)
    lib().qo_lowpass_taps(lib().qo_cutoff(frequency, sample_rate), size, _p(out))
    return out


def shift_ratio(frequency, sample_rate):
    return lib().qo_shift_ratio(frequency, sample_rate)


def shift_multipliers(ratio, ns):
    out = np.zeros((len(ns), 2), dtype=np.float32)
    c = C.c_float()
    s = C.c_float()
    for i, n in enumerate(ns):
        lib().qo_shift_multiplier(ratio, int(n), C.byref(c), C.byref(s))
        out[i] = (c.value, s.value)
    return out


def shift_apply(x, abs_off, ratio):
    out = np.array(as_c32(x), dtype=np.float32, copy=True)
    lib().qo_shift_apply(_p(out), out.shape[0], int(abs_off), ratio)
    return out


def lowpass_block(taps_, D, raw, valid=None, out_cap=None):
    raw = np.ascontiguousarray(as_c32(raw))
    valid = raw.shape[0] if valid is None else valid
    T = len(taps_)
    out_cap = (valid - T) // D if out_cap is None else out_cap
    out = np.zeros((max(out_cap, 1), 2), dtype=np.float32)
    t = np.ascontiguousarray(taps_, dtype=np.float32)
    got = lib().qo_lowpass_block(_p(t), T, D, _p(raw), valid, _p(out), out_cap)
    return got, out[:out_cap]


def fft(x):
    """restated rustfft Radix4 forward FFT of float32 pairs (n,2), n a power of two"""
    buf = np.array(as_c32(x), dtype=np.float32, copy=True)
    p = lib().qo_fft_new(buf.shape[0])
    if not p:
        raise ValueError("Radix4 requires a power-of-two length")
    lib().qo_fft_process(p, _p(buf))
    lib().qo_fft_free(p)
    return buf


def fft_twiddles(n):
    p = lib().qo_fft_new(n)
    cnt = lib().qo_fft_twiddle_count(p)
    base = lib().qo_fft_base_len(p)
    ptr = lib().qo_fft_twiddles(p)
    tw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(max(cnt, 1) * 2,)).copy()[: cnt * 2].reshape(-1, 2)
    lib().qo_fft_free(p)
    return base, tw


def dft_f64(x):
    buf = np.ascontiguousarray(as_c32(x))
    n = buf.shape[0]
    re = np.zeros(n)
    im = np.zeros(n)
    lib().qo_dft_f64(_p(buf), n, _p(re), _p(im))
    return re + 1j * im


def norm(x):
    x = np.ascontiguousarray(as_c32(x))
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().qo_norm_batch(_p(x), x.shape[0], _p(out))      # hypotf per element through the same libm the oracle links
    return out
