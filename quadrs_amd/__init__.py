"""quadrs_amd — MI355X (gfx950) engine for quadrs' IQ-stream hot path.

The product is quadrs_amd/libquadrs_hip.so (hand-written HIP kernels behind the C ABI in
include/quadrs_hip.h).  This package is the thin Python view of that ABI used by tests and
bench.py; it mirrors the reference's operator names (from / shift / lowpass / sparkfft /
bucket / gen) and error behaviour.  It never computes on the CPU: if the HIP library is
missing, importing `quadrs_amd.engine` objects raises.
"""
from . import _ffi  # noqa: F401
from .engine import (PinnedBuffer, Plan, options_from_env, plan_options, fft_norm_batch, gen, gen_device, lowpass_block, lowpass_design, shift, shift_ratio,  # noqa: F401
                     take_fft, unpack)
from ._ffi import (MODE_EXACT, MODE_FAST, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_NO_PLAN_TIME, KERNEL_SPECIALISE, MEM_DEVICE, MEM_HOST, MEM_HOST_PINNED,  # noqa: F401
                   EPI_BUCKET2_U8, EPI_CF32_BLOCKS, EPI_GLYPH_U8, EPI_NORMS_F32, FMT_CF32, FMT_CS16, FMT_CS8, FMT_CU8,  # noqa: F401
                   QuadrsError)
