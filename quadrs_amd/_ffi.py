"""ctypes binding of include/quadrs_hip.h (the C ABI).  No torch types cross this boundary:
device buffers are passed as integer addresses (e.g. tensor.data_ptr())."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QD_LIB_PATH") or os.path.join(_HERE, "libquadrs_hip.so")   # QD_LIB_PATH: diagnostic builds

OK, ERR_INVALID, ERR_PANIC, ERR_SHORT, ERR_HIP, ERR_UNSUPPORTED = range(6)
FMT_CF32, FMT_CS8, FMT_CU8, FMT_CS16 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALISE, KERNEL_NO_PLAN_TIME = 0, 1, 2, 3
MODE_EXACT, MODE_FAST = 0, 1
MAX_SHARDS = 16
EPI_NORMS_F32, EPI_GLYPH_U8, EPI_BUCKET2_U8, EPI_CF32_BLOCKS = 0, 1, 2, 3

# every symbol include/quadrs_hip.h declares
SYMBOLS = [
    "qd_last_error", "qd_version", "qd_device_count", "qd_set_device", "qd_pair_bytes", "qd_unpack",
    "qd_shift_ratio", "qd_shift", "qd_lowpass_design", "qd_lowpass_block", "qd_fft_norm_batch",
    "qd_plan_create", "qd_plan_destroy", "qd_plan_get_info", "qd_plan_get_taps", "qd_plan_src_range",
    "qd_plan_run", "qd_plan_set_timing", "qd_plan_last_kernel_ms", "qd_gen", "qd_take_fft",
    "qd_device_alloc", "qd_device_free", "qd_device_copy",
    "qd_set_stream", "qd_release_workspaces", "qd_plan_create_ex", "qd_plan_shard_info", "qd_plan_run_sharded",
    "qd_plan_run_sharded_device", "qd_plan_get_stats", "qd_host_alloc", "qd_host_free", "qd_host_register",
    "qd_host_unregister", "qd_plan_kernel_name",
]


class ChainDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("format", C.c_int32), ("sample_rate", C.c_uint64),
        ("n_samples", C.c_uint64), ("has_shift", C.c_int32), ("_pad0", C.c_int32),
        ("shift_hz", C.c_int64), ("has_lowpass", C.c_int32), ("_pad1", C.c_int32),
        ("lowpass_hz", C.c_uint64), ("decimate", C.c_uint64), ("taps", C.c_uint64),
        ("width", C.c_uint64), ("stride", C.c_uint64), ("epilogue", C.c_int32),
        ("has_range", C.c_int32), ("range_min", C.c_float), ("range_max", C.c_float),
        ("mode", C.c_int32), ("_pad2", C.c_int32),
    ]


class PlanInfo(C.Structure):
    _fields_ = [
        ("n_windows", C.c_uint64), ("decimated_len", C.c_uint64), ("out_sample_rate", C.c_uint64),
        ("out_bytes_per_window", C.c_uint64), ("raw_per_window", C.c_uint64), ("raw_step", C.c_uint64),
        ("ratio", C.c_double), ("tile_windows", C.c_uint32), ("threads", C.c_uint32),
        ("lds_bytes", C.c_uint32), ("kernel_kind", C.c_uint32), ("kernel_flags", C.c_uint32), ("_reserved", C.c_uint32),
    ]


class PlanOptions(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("kernel_policy", C.c_int32), ("nco_order", C.c_int32),
        ("copy_threads", C.c_uint32), ("chunk_bytes", C.c_uint64), ("n_shards", C.c_uint32),
        ("shard_device", C.c_int32 * MAX_SHARDS), ("tile_hint", C.c_uint32 * 8),
    ]


class ShardInfo(C.Structure):
    _fields_ = [
        ("w0", C.c_uint64), ("w1", C.c_uint64), ("own_first", C.c_uint64), ("own_count", C.c_uint64),
        ("halo", C.c_uint64), ("device", C.c_int32), ("_pad", C.c_int32),
    ]


class PlanStats(C.Structure):
    _fields_ = [
        ("wall_ms", C.c_double), ("stage_ms", C.c_double), ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64),
        ("chunks", C.c_uint32), ("_pad", C.c_uint32),
    ]


class QuadrsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"quadrs-hip status {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load the HIP extension.  There is no CPU fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python quadrs_amd/build.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7,
        # but its dependants ask for "libamdhip64.so"), so if this library were loaded first the
        # process would end up with two runtimes and torch tensors / streams could not be shared.
        # Importing torch first makes our DT_NEEDED libamdhip64.so.7 resolve to torch's copy.
        try:
            import torch  # noqa: F401
        except ImportError:      # a torch-less host (e.g. the C++ CLI) simply uses /opt/rocm's runtime
            pass
        L = C.CDLL(LIB_PATH)
        vp, u64, i64, sz, f32, f64, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_size_t, C.c_float, C.c_double, C.c_int
        sig = {
            "qd_last_error": (C.c_char_p, []),
            "qd_version": (C.c_char_p, []),
            "qd_device_count": (i32, [C.POINTER(i32)]),
            "qd_set_device": (i32, [i32]),
            "qd_pair_bytes": (u64, [i32]),
            "qd_unpack": (i32, [i32, vp, sz, vp, i32]),
            "qd_shift_ratio": (f64, [i64, u64]),
            "qd_shift": (i32, [vp, sz, u64, f64, i32]),
            "qd_lowpass_design": (i32, [u64, u64, sz, vp]),
            "qd_lowpass_block": (i32, [vp, sz, u64, vp, sz, vp, sz, C.POINTER(sz), i32]),
            "qd_fft_norm_batch": (i32, [vp, sz, sz, sz, vp, i32]),
            "qd_plan_create": (i32, [C.POINTER(ChainDesc), C.POINTER(vp)]),
            "qd_plan_destroy": (i32, [vp]),
            "qd_plan_get_info": (i32, [vp, C.POINTER(PlanInfo)]),
            "qd_plan_get_taps": (i32, [vp, vp, sz]),
            "qd_plan_kernel_name": (i32, [vp, C.c_char_p, sz]),
            "qd_plan_src_range": (i32, [vp, u64, u64, C.POINTER(u64), C.POINTER(u64)]),
            "qd_plan_run": (i32, [vp, vp, i32, u64, u64, u64, u64, vp, i32, vp]),
            "qd_plan_set_timing": (i32, [vp, i32]),
            "qd_plan_last_kernel_ms": (i32, [vp, C.POINTER(f32)]),
            "qd_gen": (i32, [vp, sz, u64, u64, sz, vp, i32]),
            "qd_device_alloc": (i32, [sz, C.POINTER(C.c_void_p)]),
            "qd_device_free": (i32, [vp]),
            "qd_device_copy": (i32, [vp, i32, vp, i32, sz]),
            "qd_take_fft": (i32, [vp, u64, sz, u64, i32, u64, u64, sz, i32, sz, vp, i32]),
            "qd_set_stream": (i32, [vp]),
            "qd_release_workspaces": (i32, []),
            "qd_plan_create_ex": (i32, [C.POINTER(ChainDesc), C.POINTER(PlanOptions), C.POINTER(vp)]),
            "qd_plan_shard_info": (i32, [vp, C.c_uint32, C.POINTER(ShardInfo)]),
            "qd_plan_run_sharded": (i32, [vp, vp, i32, vp, i32]),
            "qd_plan_run_sharded_device": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32]),
            "qd_plan_get_stats": (i32, [vp, C.POINTER(PlanStats)]),
            "qd_host_alloc": (i32, [sz, C.POINTER(vp)]),
            "qd_host_free": (i32, [vp]),
            "qd_host_register": (i32, [vp, sz]),
            "qd_host_unregister": (i32, [vp]),
        }
        for name, (res, args) in sig.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                # only a diagnostic library named by QD_LIB_PATH (an older build kept for a before / after timing) may lack an entry point
                if os.environ.get("QD_LIB_PATH") and name in ("qd_plan_kernel_name",):
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != OK:
        raise QuadrsError(code, lib().qd_last_error().decode(errors="replace"))
