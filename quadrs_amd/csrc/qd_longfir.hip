// qd_longfir.hip — shape-specialised chain kernels for FIR-dominated shapes (T/D >> 1, overlapping windows).
//
// Why a separate translation unit: these kernels spend most of a tile in the per-output accumulate chain
// (acc += x * h, ascending taps, separately rounded: src/filter.rs:107-124).  With the SLP vectorizer on,
// hipcc fuses the re/im halves into one v_pk_mul_f32 / v_pk_add_f32 chain per output; on gfx950 a packed
// op costs as much as two scalar ones but the single dependent chain has twice the latency per tap, and
// the tap operands need extra register shuffles.  Scalar chains measured 12% faster on this shape
// (DESIGN.md section 7), while the short-filter kernels in quadrs_hip.hip lose 50% without SLP (their NCO /
// FFT code packs well).  So: same source, different flag, per shape.  build.py compiles this file with
// -fno-slp-vectorize.
//
// Tiling: the FIR phase of a tile is latency-bound (one wave walks all T taps however few outputs the
// tile has), so these shapes take the largest tile LDS allows: 27 windows of 64 = 480 FIR outputs, one ~156 KiB
// tile per CU.  1024 threads: the FIR keeps 8 of the 16 waves busy, but phase 1 (unpack + f64 NCO over 15 792 samples,
// 40 % of the tile) runs on all 16 (31.1 ms vs 32.3 ms with 512 threads and a 256-VGPR budget).  LDS rows are 16-byte aligned
// (PAD 2): the tap loop reads sample pairs with conflict-free ds_read_b128 instead of the ds_read2_b64 hipcc forms
// from 8-byte reads (half the bytes per clock on gfx950); that costs one window of tile (27 instead of 28 fit) and
// is worth 4 % once the loop itself is tight (33.7 -> 32.3 ms on cfg3).
#include "qd_registry.h"

namespace qd {

static const FixedEntry kLongFir[] = {
    // README.md:90-94 / configs[2]  "lowpass -power 200 -decimate 32 200000 | sparkfft -width 64 -stride 16"
    // cf32 input (the README's own FSK example file, BASELINE configs[4]): the three-stage kernel in quadrs_hip.hip (k_chain_pipe3)
#ifndef QD_DEV_FAST
    // cs8 input (HackRF): 4096-sample rows, 5 per tile, whole-tile register prefetch
    QD_FIXED_NTF(1, 1, 64, 16, 32, 400, 27, 5, true, 4, 1024, 8, 1, 2, 32, "cfg3"),
    QD_FIXED_NTF(1, 2, 64, 16, 32, 400, 27, 5, true, 4, 1024, 8, 1, 2, 32, "cfg3"),
#endif
};

const FixedEntry *longfir_entries(int *count) {
    *count = (int)(sizeof kLongFir / sizeof kLongFir[0]);
    return kLongFir;
}

}  // namespace qd
