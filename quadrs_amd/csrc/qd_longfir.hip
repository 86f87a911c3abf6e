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
// tile has), so these shapes take the largest tile LDS allows: 28 windows of 64 = 496 FIR outputs for 512
// lanes, one 156 KiB tile per CU, 256-VGPR budget (no spills in the tap loop).  The 16-byte-aligned LDS layout
// (PAD 2, ds_read_b128 pairs) was measured here too: it costs a window of tile (27 instead of 28 fit) and the
// tap loop is not LDS-throughput-bound (19.1k vs 20.3k cycles for 480 vs 496 outputs), so PAD stays 1.
#include "qd_registry.h"

namespace qd {

static const FixedEntry kLongFir[] = {
    // README.md:90-94 / configs[2]  "lowpass -power 200 -decimate 32 200000 | sparkfft -width 64 -stride 16"
    // cf32 input (the README's own FSK example file): 1024-sample rows, 17 per tile -> chunked prefetch
    QD_FIXED_NT(0, 1, 64, 16, 32, 400, 28, 4, false, 2, 512, 8, 1, 1, "fsk5"),
    QD_FIXED_NT(0, 2, 64, 16, 32, 400, 28, 4, false, 2, 512, 8, 1, 1, "fsk5"),
#ifndef QD_DEV_FAST
    // cs8 input (HackRF): 2048-sample rows, 9 per tile, whole-tile register prefetch
    QD_FIXED_NT(1, 1, 64, 16, 32, 400, 28, 9, true, 2, 512, 8, 1, 1, "cfg3"),
    QD_FIXED_NT(1, 2, 64, 16, 32, 400, 28, 9, true, 2, 512, 8, 1, 1, "cfg3"),
#endif
};

const FixedEntry *longfir_entries(int *count) {
    *count = (int)(sizeof kLongFir / sizeof kLongFir[0]);
    return kLongFir;
}

}  // namespace qd
