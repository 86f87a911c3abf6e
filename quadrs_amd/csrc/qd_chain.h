// qd_chain.h — the fused chain kernel:  unpack -> shift -> lowpass(FIR+decimate) -> FFT -> |X| -> epilogue
//
// One workgroup (256 threads) owns a *tile* of G consecutive FFT windows of the sink's loop
// (spark_fft, src/fft.rs:28-65).  Per tile:
//   phase 1  stream the tile's contiguous raw range from HBM (16-byte coalesced loads, prefetched
//            into registers one tile / several rows ahead), unpack, multiply by the NCO, park the
//            shifted cf32 samples in LDS (row-padded so that the FIR's stride-D lane pattern is
//            bank-conflict free);
//   phase 2  FIR+decimate, one lane per decimated output, taps ascending, separately rounded
//            mul/add — the reference's summation order (src/filter.rs:111-121), including its
//            per-read_at tail truncation (jmax), results scattered into the FFT buffer in
//            rustfft's digit-reversed order;
//   phase 3  W-point FFT in LDS with rustfft's scalar Radix4 structure (base butterfly +
//            radix-4 DIT layers);
//   phase 4  fftshift + hypot (+ glyph / bucket) and a coalesced store.
// Decimated samples never leave the CU.  Windows are independent units, so tiles need no
// inter-workgroup communication; the grid is sized to the chip and strides over tiles.
//
// The kernel is written once against a geometry policy: FixedGeo<W,S,D,T,G> turns every shape
// parameter into a compile-time constant (LDS offsets become immediates, the FIR unrolls, index
// math folds away) and is instantiated for common chain shapes; DynGeo reads the same quantities
// from the launch parameters and serves every other shape.
#pragma once

#include "qd_device.h"

namespace qd {

// In-kernel phase stamps (cdna_hip_programming.md §7): diagnostic build only, never in the shipped .so.
#ifdef QD_STAMP
#define QD_STAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_row[4] = {0, 0, 0, 0}; unsigned st_tiles = 0;
#define QD_STAMP_ROW(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_row[k] += t_ - st_prev; st_acc[0] += t_ - st_prev; st_prev = t_; } while (0)
#define QD_STAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory"); } while (0)
#define QD_STAMP_AT(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define QD_STAMP_FLUSH() do { if (P.stamps && (threadIdx.x & 63) == 0 && threadIdx.x < 1024) { unsigned w_ = threadIdx.x >> 6; for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&P.stamps[w_ * 8 + k_], st_acc[k_]); if (w_ == 1) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&P.stamps[129 + k_], st_row[k_]); if (w_ == 0) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&P.stamps[133 + k_], st_row[k_]); if (w_ == 0) atomicAdd(&P.stamps[128], (unsigned long long)st_tiles); } } while (0)
#define QD_STAMP_TILE() do { ++st_tiles; } while (0)
#else
#define QD_STAMP_DECL
#define QD_STAMP_START()
#define QD_STAMP_AT(k)
#define QD_STAMP_FLUSH()
#define QD_STAMP_TILE()
#define QD_STAMP_ROW(k)
#endif

constexpr int kThreads = 256;

// Plan-time builds may bake the plan's filter into the code (FixedGeo FLAGS_ bit 1): the host puts
// `#define QD_BAKED_TAPS_LIST 0x1.8p-7f, ...` (the designed taps, exact hex floats) in front of this header.
#ifdef QD_BAKED_TAPS_LIST
constexpr float kBakedTapTable[] = {QD_BAKED_TAPS_LIST};
#else
constexpr float kBakedTapTable[1] = {0.f};
#endif

// Timing-only ablation bits (ChainParams::dbg) exist in development builds (-DQD_DEVELOP: libquadrs_hip_dev.so, the
// probe scripts) only; in the shipped library the tests compile to nothing.
#ifdef QD_DEVELOP
#define QD_DBG(P, bit) (((P).dbg & (bit)) != 0)
#else
#define QD_DBG(P, bit) false
#endif

// Wave-uniform read-only tables (taps, row bases) are read through the constant address space
// so that hipcc emits scalar loads (s_load_dwordx8) instead of one vector load per lane.
typedef const float __attribute__((address_space(4))) *const_f32_p;
typedef const double __attribute__((address_space(4))) *const_f64_p;

// samples per lane per row-load, by format
template <int FMT> struct FmtTraits;
template <> struct FmtTraits<0> { static constexpr int BPS = 8; static constexpr int SPL = 2; using Vec = uint4; };  // cf32: 16 B / lane
template <> struct FmtTraits<1> { static constexpr int BPS = 2; static constexpr int SPL = 4; using Vec = uint2; };  // cs8 :  8 B / lane
template <> struct FmtTraits<2> { static constexpr int BPS = 2; static constexpr int SPL = 4; using Vec = uint2; };  // cu8 :  8 B / lane
template <> struct FmtTraits<3> { static constexpr int BPS = 4; static constexpr int SPL = 4; using Vec = uint4; };  // cs16: 16 B / lane

struct ChainParams {
    // ---- per launch
    const uint8_t *src;        // raw bytes of sample src_first
    uint64_t src_first;
    uint64_t src_count;
    uint64_t first_window;     // of this launch
    uint64_t n_windows;
    uint64_t out_window0;      // window index that maps to out[0]
    void *out;
    const RowBase *rowtab;     // indexed by absolute row - rowtab_row0
    uint64_t rowtab_row0;
    const double2 *jtab;       // ROW entries (cos, sin)(fl(j*ratio))
    const float *taps;
    const float2 *tw;          // radix-4 layer twiddles, bottom layer first
    double ratio;
    float rmin, rmax;
    float root2;
    float2 tw16_1, tw16_2, tw16_3;
    uint32_t epi;              // qd_epilogue
    float gstep;               // glyph epilogue: (rmax - rmin) / 7.0f, computed on the host (see glyph_code)
    float rgstep;              // ... and RN(1 / gstep): the short form's multiplier
    uint32_t dbg;              // development builds (-DQD_DEVELOP) only: timing-only ablation bits (1 NCO, 2 FIR, 4 FFT, 8 hypot, 16 output store, 32 LDS staging); else only the never-true liveness sentinel reads it
    unsigned long long *stamps; // diagnostic builds (-DQD_STAMP) only: per-phase cycle sums, else unused
    unsigned long long *work;   // dynamic tile queue: 8 per-XCD-group counters + 1 arrival counter, 16 words (128 B) apart, all zero
                                // at launch (the last workgroup to finish zeroes them again); NULL: static strided walk
    const uint64_t *row_offsets; // take_fft (src/ffts.rs:59-60): window w starts at row_offsets[w] (generic kernels, G = 1)
    const float *window;         // take_fft windowing (src/ffts.rs:64-68): sample k of a window is scaled by window[k]
    uint32_t blk_len;            // length B of the read_at block the truncation is relative to (== W except QD_EPI_CF32_BLOCKS)
    uint32_t blk_sub_mask;       // QD_EPI_CF32_BLOCKS: (B / W) - 1, sub-windows per block minus one
    uint32_t tile_extra;         // QD_EPI_CF32_BLOCKS: max(0, c - D) more raw samples per tile (a tile that ends inside a
                                 // block has untruncated outputs whose taps reach past W*D + T); 0 otherwise
    // ---- geometry (only DynGeo reads these; FixedGeo has them as constants)
    uint32_t W, logW, S, D, T, G;
    uint32_t Dp;               // LDS row pitch: D + 1 if D even else D
    uint32_t dmagic;           // floor(2^32 / D) + 1  (exact m / D for m*D < 2^32)
    uint32_t dshift;           // log2(D) if D is a power of two, else 0xffffffff
    uint32_t a0, b0;           // c = T - T/2 = a0*D + b0
    uint32_t T_fast;           // min(T, D + T/2): every output's jmax is >= this
    uint32_t a1, b1;           // c + T_fast = a1*D + b1
    uint32_t base_len, log_base, layers;   // rustfft Radix4 plan: W = base_len * 4^layers
    uint32_t lds_raw_elems;    // float2 capacity of the raw tile
    uint32_t lds_dyn;          // dynamic LDS bytes of this launch: the kernels with a layout of their own (three-stage kernels) check it
                               // against their compile-time need and do nothing if the host's restatement of the layout fell short
    uint32_t out_row_stride;   // wave-local kernels (plan-time builds): output rows between consecutive windows of THIS launch (0 / 1: contiguous).
                               // Overlapping windows without a lowpass run as W / S interleaved launches — launch phi takes the windows
                               // phi, phi + R, phi + 2R ... (R = W / S), which lie side by side in the stream shifted by phi * S samples
};

// The glyph cell of a norm: the short form (qd_device.h) in the shape-specialised kernels; the runtime-geometry kernels keep the literal
// division — they have no scalar register to spare for one more kernel argument (test_builtin_kernels_do_not_spill bounds their spills).
template <class GeoT>
__device__ __forceinline__ uint8_t glyph_of(const ChainParams &P, float nm) {
    if constexpr (GeoT::kFixed) return glyph_code(nm, P.rmin, P.rmax, P.gstep, P.rgstep);
    else return glyph_code_ieee(nm, P.rmin, P.rmax, P.gstep);
}

// ---------------------------------------------------------------- geometry policies

constexpr uint32_t ct_log2(uint32_t v) { uint32_t l = 0; while ((1u << l) < v) ++l; return l; }
constexpr bool ct_pow2(uint32_t v) { return v && !(v & (v - 1)); }
constexpr uint32_t ct_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

// LDS float2 elements the raw tile needs (same formula as the host's lds_for())
constexpr uint32_t ct_raw_elems(uint32_t W, uint32_t S, uint32_t D, uint32_t T, uint32_t G, uint32_t pad_per_row = 1) {
    uint32_t tile_raw = (G - 1) * S * D + W * D + T;
    uint32_t pad = (D % 2 == 0) ? pad_per_row * (tile_raw / D + 1) : 0;
    uint32_t elems = tile_raw + pad + 1;
    uint32_t min_elems = G * W / 2 + 1;
    if (elems < min_elems) elems = min_elems;
    return (elems + 1) & ~1u;
}

// planar raw tile (FixedGeo FLAGS_ bit 0): floats per plane for a tile of tile_raw samples, rows of D floats at pitch DpP
constexpr uint32_t ct_planar_pitch(uint32_t D) { return ((D / 4) % 2 == 1) ? D : D + 4; }     // 16-byte aligned, pitch/4 odd
constexpr uint32_t ct_plane_floats(uint32_t W, uint32_t S, uint32_t D, uint32_t T, uint32_t G) {
    const uint32_t tile_raw = (G - 1) * S * D + W * D + T;
    return ((tile_raw / D + 1) * ct_planar_pitch(D) + 7) & ~7u;
}

constexpr uint32_t kGeoPlanar = 1, kGeoBakedTaps = 2, kGeoNoSplit = 4, kGeoFastP1 = 8, kGeoPackedSpan = 16, kGeoUnrolledFir = 32,
                   kGeoDeferFft = 64, kGeoPackedTile = 128, kGeoNtLoads = 256;   // FixedGeo FLAGS_ bits
// cache policy (buffer-load aux operand) of the phase-1 stream loads: bit 8 -> nt, bits 11 / 12 -> sc0 / sc1
constexpr uint32_t kGeoNtInner = 65536;      // FLAGS_ bit 16 (with bit 8): rows a neighbouring tile reads too (the first and the last of a row-aligned tile) keep the default policy
constexpr uint32_t kGeoFastFma = 16384;      // FLAGS_ bit 14: QD_MODE_FAST — the packed FIRs fuse multiply and add (v_pk_fma_f32), one rounding per tap
constexpr uint32_t kGeoHalfTile = 8192;      // FLAGS_ bit 13: the tile buffer holds HALF a window's FIR input, two passes per window (see FixedGeo::kHalfTile)
constexpr int ct_load_aux(uint32_t flags) { return ((flags & 256u) ? 2 : 0) | ((flags & 2048u) ? 1 : 0) | ((flags & 4096u) ? 16 : 0); }
typedef float v2f __attribute__((ext_vector_type(2)));      // operand type of the v_pk_*_f32 instructions

template <uint32_t W_, uint32_t S_, uint32_t D_, uint32_t T_, uint32_t G_, uint32_t FIRB_ = 8, uint32_t FIRR_ = 1, uint32_t PAD_ = 1, uint32_t BATCH_ = 1,
          uint32_t FLAGS_ = 0>
struct FixedGeo {
    static constexpr bool kFixed = true;
    static constexpr uint32_t kFlags = FLAGS_;
    // Planar raw tile: the shifted samples are parked as two f32 planes (re / im) instead of interleaved pairs, rows of D floats
    // at a 16-byte aligned pitch DpP with DpP/4 odd.  The component-split FIR then reads FOUR taps of its component with one
    // conflict-free ds_read_b128 (256 B/clk) instead of two with a ds_read2_b32 (128 B/clk): half the LDS-array cycles and
    // half the LDS instructions of the dominant loop.
    static constexpr uint32_t DpP = ct_planar_pitch(D_);
    static constexpr uint32_t plane_floats = ct_plane_floats(W_, S_, D_, T_, G_);
    static constexpr bool planar_geometry = (FLAGS_ & kGeoPlanar) && ct_pow2(D_) && D_ % 8 == 0 && T_ % 4 == 0 && ((T_ - T_ / 2) % D_) % 4 == 0 && T_ >= 64 &&
                                            ct_pow2(G_ * W_);
    // Baked taps (plan-time builds only): the filter is a compile-time table, so every tap is an immediate operand of its
    // multiply — no LDS reads, no registers, no lgkmcnt traffic for the taps at all.
    static constexpr bool baked_request = (FLAGS_ & kGeoBakedTaps) != 0;
    // FFT batching: the decimated windows of BATCH_ consecutive tiles of a workgroup are parked in LDS and transformed
    // together.  The FFT + epilogue of ONE 128-point window keeps 16-32 of 256 lanes busy between three barriers and is
    // pure latency; B windows at once cost the same latency for B times the work.
    static constexpr uint32_t kBatch = BATCH_ ? BATCH_ : 1;
    static constexpr uint32_t kFirBlock = FIRB_;   // taps per software-pipelined FIR block (register budget knob)
    // outputs per lane in the FIR (register tiling): each LDS sample read feeds FIRR_ accumulators.
    // Needs 8-aligned geometry; falls back to 1 otherwise.
    // The straight-line packed two-output form (fir_tiled2_pk, FLAGS_ bit 7) walks 4-sample blocks with compile-time offsets
    // and needs 4-aligned geometry only (T = 200: c = 100, T/2 = 100).
    static constexpr bool kFirTile4 = FIRR_ == 2 && (FLAGS_ & kGeoPackedTile) && PAD_ == 2 && D_ % 4 == 0 && (T_ - T_ / 2) % 4 == 0 && T_ % 4 == 0 &&
                                      (T_ / 2) % 4 == 0 && W_ % 2 == 0 && S_ % 2 == 0 && ct_pow2(D_) && D_ / 4 <= 8 && T_ > D_ + 16 && !(T_ > 0 && S_ < W_);
    static constexpr uint32_t kFirTile = kFirTile4 ? 2u :
                                         (FIRR_ > 1 && D_ % 8 == 0 && ((T_ - T_ / 2) % D_) % 8 == 0 && T_ % 8 == 0 && (T_ / 2) % 8 == 0 &&
                                          W_ % FIRR_ == 0 && S_ % FIRR_ == 0 && ct_pow2(D_) && ct_pow2(FIRR_) &&
                                          T_ > (FIRR_ - 1) * D_ + 8) ? FIRR_ : 1;      // at least three interior 4-sample blocks
    // LDS pad period: one pad element per PD samples.  PD = D for the lane-per-output FIR (lane stride D + 1:
    // odd, conflict-free ds_read_b64); the register-tiled FIR strides lanes by kFirTile rows, so it pads once
    // per kFirTile * D samples to keep the lane stride odd.
    static constexpr uint32_t PD = D_ * kFirTile;
    static constexpr uint32_t pshift = ct_pow2(PD) ? ct_log2(PD) : 0xffffffffu;
    __device__ __forceinline__ explicit FixedGeo(const ChainParams &) {}
    static constexpr uint32_t W = W_, S = S_, D = D_, T = T_, G = G_;
    static constexpr uint32_t G_ct = G_, W_ct = W_;
    static constexpr uint32_t logW = ct_log2(W_);
    // LDS pad elements per pad period.  1: odd lane stride, conflict-free single ds_read_b64 — but hipcc merges
    // neighbouring reads into ds_read2_b64, which gfx950 serves at half the bytes per clock.  2 (even D, even
    // first column): rows stay 16-byte aligned and the lane stride is 4*odd banks, so the FIR reads sample
    // PAIRS with conflict-free ds_read_b128 at the full 256 B/clk (MI355X_MICROARCH.md, LDS table).
    static constexpr uint32_t kPad = (D_ % 2 == 0) ? ((PAD_ == 2 && (T_ - T_ / 2) % 2 == 0) ? 2u : 1u) : 0u;
    static constexpr uint32_t Dp = D_ + kPad;
    static constexpr uint32_t dshift = ct_pow2(D_) ? ct_log2(D_) : 0xffffffffu;
    static constexpr uint32_t dmagic = D_ > 1 ? (uint32_t)((1ull << 32) / D_ + 1) : 0u;
    static constexpr uint32_t c = T_ - T_ / 2;
    static constexpr uint32_t a0 = c / D_, b0 = c % D_;
    static constexpr uint32_t T_fast = ct_min(T_, D_ + T_ / 2);
    static constexpr uint32_t a1 = (c + T_fast) / D_, b1 = (c + T_fast) % D_;
    static constexpr uint32_t log_base = logW <= 3 ? logW : ((logW & 1) ? 3u : 4u);
    static constexpr uint32_t base_len = 1u << log_base;
    static constexpr uint32_t layers = (logW - log_base) / 2;
    // Half-window tiles (FLAGS_ bit 13; one long window per tile, non-overlapping): the raw buffer holds the input of HALF the
    // window's outputs (c + (W/2 - 1) D + T samples) and a window is filtered in two passes — phase 1 + FIR of outputs [0, W/2),
    // then of [W/2, W) — into one FFT slot.  The tile is half as large, so TWO workgroups fit on a CU where one did (cfg4: 97 KiB ->
    // 62 KiB), and one workgroup's phase 1 and barriers run while the other's FIR has the vector units: the 8 k serial cycles in
    // front of every 21 k-cycle FIR of the one-workgroup form overlap.  Every output sees the same samples, taps and order.
    static constexpr bool kHalfTile = (FLAGS_ & kGeoHalfTile) && G_ == 1 && !(T_ > 0 && S_ < W_) && W_ % 4 == 0 && T_ > 0;
    static constexpr uint32_t kHalfOut = W_ / 2;
    static constexpr uint32_t kHalfRaw = (T_ - T_ / 2) + (kHalfOut - 1) * D_ + T_;
    static constexpr uint32_t ct_half_elems() {
        const uint32_t pad = (D_ % 2 == 0) ? (kPad ? kPad : 1) * (kHalfRaw / D_ + 1) : 0;
        uint32_t elems = kHalfRaw + pad + 1;
        const uint32_t min_elems = G_ * W_ / 2 + 1;
        if (elems < min_elems) elems = min_elems;
        return (elems + 1) & ~1u;
    }
    static constexpr uint32_t lds_raw_elems_std = kHalfTile ? ct_half_elems() : ct_raw_elems(W_, S_, D_, T_, G_, kPad ? kPad : 1);
    static constexpr uint32_t kNtrunc = c ? (c + D_ - 1) / D_ - 1 : 0;
    static constexpr bool kShared = T_ > 0 && S_ < W_ && kNtrunc <= S_;     // shared-FIR mode (see phase 2)
    // component-split FIR (fir_comp): mid-length filters whose tile leaves at least half the lanes without an output
    static constexpr bool split_ok(uint32_t nt) {
        return !(FLAGS_ & kGeoNoSplit) && !kShared && kFirTile == 1 && kPad != 2 && D_ % 8 == 0 && T_ >= 64 && 2u * G_ * W_ <= nt;
    }
    // register-tiled kernels: spare waves take the truncated tails (fir_prefix) when main lanes fill whole waves
    static constexpr bool helper_ok(uint32_t nt) {
        return kFirTile > 1 && !kShared && (G_ * W_ / kFirTile) % 64 == 0 && G_ * W_ / kFirTile + G_ * kNtrunc <= nt &&
               (T_ / 2) % 8 == 0 && kNtrunc > 0 && kNtrunc < W_;
    }
    static constexpr bool split_ok_shared(uint32_t nt) {      // same, shared-FIR mode: (G-1)*S + W outputs per tile
        return kShared && kFirTile == 1 && kPad != 2 && D_ % 8 == 0 && T_ >= 64 && 2u * ((G_ - 1) * S_ + W_) <= nt;
    }
    // the planar layout serves the component-split FIR of non-overlapping-window tiles (every 256-thread shape that asks for it)
    // packed lane-per-output FIR (fir_pair): FLAGS_ bit 2 on a 16-byte-row tile
    static constexpr bool kPairFir = (FLAGS_ & kGeoNoSplit) && !kShared && kFirTile == 1 && kPad == 2 && T_ % 4 == 0 && ((T_ - T_ / 2) % D_) % 2 == 0 &&
                                     D_ % 4 == 0 && T_ >= 32;
    static constexpr bool kUnrolledShared = (FLAGS_ & kGeoUnrolledFir) && kShared && kFirTile == 1 && kPad == 2 && T_ % 4 == 0 &&
                                            ((T_ - T_ / 2) % D_) % 2 == 0 && D_ % 4 == 0 && T_ >= 32;
    // register-tiled FIR as straight-line packed code with in-chain snapshots (fir_tiled2_pk): FLAGS_ bit 7, two outputs per lane
    static constexpr bool kPackedTile = (FLAGS_ & kGeoPackedTile) && !kShared && kFirTile == 2 && kPad == 2 && (T_ / 2) % 4 == 0 && D_ / 4 <= 8;
    static constexpr bool kPlanar = planar_geometry && split_ok(256u);
    static constexpr bool kBakedTaps = baked_request && kPlanar;       // only the planar FIR takes its taps as immediates
    static constexpr uint32_t lds_raw_elems = kPlanar ? plane_floats : lds_raw_elems_std;      // float2 elements
};

struct DynGeo {
    static constexpr bool kFixed = false;
    static constexpr uint32_t kBatch = 1;
    static constexpr uint32_t kFlags = 0, G_ct = 1, W_ct = 1;
    static constexpr bool kPlanar = false, kBakedTaps = false, kPairFir = false, kUnrolledShared = false, kPackedTile = false, kHalfTile = false;
    static constexpr uint32_t kHalfOut = 0, kHalfRaw = 0;
    static constexpr uint32_t DpP = 0, plane_floats = 0;
    static constexpr bool kShared = false;
    static constexpr uint32_t kFirTile = 1;
    static constexpr bool split_ok(uint32_t) { return false; }
    static constexpr bool helper_ok(uint32_t) { return false; }
    static constexpr bool split_ok_shared(uint32_t) { return false; }
    uint32_t W, S, D, T, G, logW, Dp, kPad, dshift, dmagic, PD, pshift, a0, b0, T_fast, a1, b1, log_base, base_len, layers, lds_raw_elems;
    __device__ __forceinline__ explicit DynGeo(const ChainParams &P)
        : W(P.W), S(P.S), D(P.D), T(P.T), G(P.G), logW(P.logW), Dp(P.Dp), kPad(P.Dp - P.D), dshift(P.dshift), dmagic(P.dmagic), PD(P.D), pshift(P.dshift),
          a0(P.a0), b0(P.b0), T_fast(P.T_fast), a1(P.a1), b1(P.b1), log_base(P.log_base), base_len(P.base_len),
          layers(P.layers), lds_raw_elems(P.lds_raw_elems) {}
};

// ---------------------------------------------------------------- phase-1 helpers
// A "row" is kThreads*SPL consecutive samples starting at an absolute multiple of ROW.

struct TileGeo {            // everything wave-uniform
    uint64_t w0, n_start;
    uint64_t r0;            // first absolute row touching the tile
    int64_t row0_off;       // byte offset of row r0 inside the slab (may be negative)
    int32_t slab_lo0, slab_hi0;   // legal byte offsets relative to row r0 (slab bounds, saturated to +-2^30)
    uint32_t n_rows;        // rows touching the tile
    uint32_t g_cnt, tile_raw;
    int32_t rel0;           // r0*ROW - n_start  (in (-ROW, 0])
    bool valid;
};

template <int FMT, int NT, class GeoT>
__device__ __forceinline__ TileGeo tile_geo(const ChainParams &P, const GeoT &geo, uint64_t tile, uint64_t n_tiles) {
    constexpr uint32_t ROW = NT * FmtTraits<FMT>::SPL;
    constexpr int BPS = FmtTraits<FMT>::BPS;
    TileGeo g;
    g.valid = tile < n_tiles;
    g.w0 = P.first_window + tile * geo.G;
    const uint64_t left = P.first_window + P.n_windows - g.w0;
    g.g_cnt = (!g.valid) ? 0u : (left < geo.G ? (uint32_t)left : geo.G);
    g.n_start = g.w0 * ((uint64_t)geo.S * geo.D);                    // LowPass reads inner at off*D (src/filter.rs:71)
    if constexpr (!GeoT::kFixed) {
        if (P.row_offsets && g.valid) g.n_start = P.row_offsets[g.w0 - P.out_window0];   // irregular rows (take_fft), uniform load
    }
    g.tile_raw = g.valid ? (g.g_cnt - 1) * geo.S * geo.D + geo.W * geo.D + geo.T : 0u;   // B*D + T (src/filter.rs:68)
    if constexpr (!GeoT::kFixed) { if (g.valid) g.tile_raw += P.tile_extra; }
    g.r0 = g.n_start / ROW;
    g.rel0 = (int32_t)(int64_t)(g.r0 * ROW - g.n_start);
    g.n_rows = (uint32_t)((g.tile_raw - g.rel0 + ROW - 1) / ROW);
    g.row0_off = ((int64_t)(g.r0 * ROW) - (int64_t)P.src_first) * BPS;
    constexpr int64_t BIG = 1ll << 30;
    constexpr int32_t VECB = FmtTraits<FMT>::SPL * BPS;
    int64_t lo = -g.row0_off, hi = (int64_t)P.src_count * BPS - VECB - g.row0_off;
    g.slab_lo0 = (int32_t)(lo < -BIG ? -BIG : (lo > BIG ? BIG : lo));
    g.slab_hi0 = (int32_t)(hi < -BIG ? -BIG : (hi > BIG ? BIG : hi));
    return g;
}

// Issue this lane's load for row i of the tile (the result stays in flight until first use).
// ALIGNED: ONE unconditional vector load per lane.  The lane offset is clamped into the slab with two
// 32-bit min/max against wave-uniform bounds, so lanes outside the slab (their data is never used)
// stay in bounds, there is no branch, and the load writes its destination register directly —
// nothing forces an early s_waitcnt.  The host never hands this path a vector that straddles the
// slab end.  !ALIGNED (slab start not vector aligned): per-sample loads, correctness path.
template <int FMT, int NT, bool ALIGNED, bool NTLOAD = false>
__device__ __forceinline__ typename FmtTraits<FMT>::Vec fetch_row(const ChainParams &P, const TileGeo &g, uint32_t i,
                                                                    uint32_t tid) {
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    constexpr int SPL = FT::SPL, BPS = FT::BPS;
    constexpr int32_t ROWB = NT * SPL * BPS, VECB = SPL * BPS;
    if constexpr (ALIGNED) {
        const int64_t row_off = g.row0_off + (int64_t)i * ROWB;                                    // uniform
        // Legal lane offsets relative to this row (all 32-bit, wave-uniform): inside the slab AND
        // inside the tile, so lanes past the tile's edge collapse onto its first / last needed
        // vector (one cache line instead of a row of never-used bytes).
        constexpr int32_t LOGV = FMT == 0 ? 4 : (FMT == 3 ? 4 : 3);                                // log2(VECB)
        const int32_t rel_i = g.rel0 + (int32_t)(i * (NT * SPL));                                  // row start - tile start
        const int32_t tlo = ((-rel_i * BPS) >> LOGV) << LOGV;                                      // vector holding the first tile byte
        const int32_t thi = (((-rel_i + (int32_t)g.tile_raw) * BPS - 1) >> LOGV) << LOGV;          // ... the last tile byte
        int32_t lo_rel = g.slab_lo0 - (int32_t)(i * ROWB), hi_rel = g.slab_hi0 - (int32_t)(i * ROWB);
        lo_rel = lo_rel > tlo ? lo_rel : tlo;
        hi_rel = hi_rel < thi ? hi_rel : thi;
        lo_rel = lo_rel < 0 ? 0 : (lo_rel > ROWB ? ROWB : lo_rel);
        hi_rel = hi_rel > ROWB ? ROWB : (hi_rel < -ROWB ? -ROWB : hi_rel);
        int32_t t = (int32_t)(tid * VECB);
        t = t < lo_rel ? lo_rel : t;
        t = t > hi_rel ? hi_rel : t;
        const uint8_t *rowp = P.src + row_off;                                                     // uniform base
        if constexpr (NTLOAD) {                       // the stream is read once: non-temporal policy (FixedGeo FLAGS_ bit 8)
            typedef unsigned vnt4 __attribute__((ext_vector_type(4)));
            typedef unsigned vnt2 __attribute__((ext_vector_type(2)));
            Vec v;
            if constexpr (sizeof(Vec) == 16) { const vnt4 w = __builtin_nontemporal_load(reinterpret_cast<const vnt4 *>(rowp + t)); v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w; }
            else { const vnt2 w = __builtin_nontemporal_load(reinterpret_cast<const vnt2 *>(rowp + t)); v.x = w.x; v.y = w.y; }
            return v;
        }
        return *reinterpret_cast<const Vec *>(rowp + t);
    } else {
        const int32_t m0 = g.rel0 + (int32_t)(i * (NT * SPL)) + (int32_t)(tid * SPL);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const int32_t m = m0 + u;
            if (m >= 0 && m < (int32_t)g.tile_raw) {        // samples inside the tile are inside the slab
                const uint8_t *sp = P.src + (g.n_start + (uint32_t)m - P.src_first) * BPS;
                if constexpr (FMT == 0) { uint2 t = *reinterpret_cast<const uint2 *>(sp); w[2 * u] = t.x; w[2 * u + 1] = t.y; }
                else if constexpr (FMT == 3) { w[u] = *reinterpret_cast<const uint32_t *>(sp); }
                else { uint32_t t = *reinterpret_cast<const uint16_t *>(sp); w[u >> 1] |= t << (16 * (u & 1)); }
            }
        }
        Vec v{};
        if constexpr (sizeof(Vec) == 16) { v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3]; }
        else { v.x = w[0]; v.y = w[1]; }
        return v;
    }
}

// The row table is streamed (32 B per row) and read with scalar loads one row ahead; without help
// every other row base is a cold HBM miss through the scalar cache (~1000+ cycles).  So a few lanes
// touch the NEXT tile's entries with a vector load a whole tile early: the lines are then in this
// XCD's L2 when the scalar loads come.  The value is only summed into a never-stored register.
template <bool HAS_SHIFT>
__device__ __forceinline__ double prefetch_rowtab(const ChainParams &P, const TileGeo &ng, uint32_t tid) {
    double v = 0.0;
    if constexpr (HAS_SHIFT) {
        // A plain load, unconditional (lanes past the last row re-touch it), index opaque so it stays inside the tile
        // loop.  NOT volatile: hipcc turns a volatile load into `flat_load ... sc0 sc1` + `s_waitcnt vmcnt(0)`, i.e. a
        // full drain of the next tile's prefetch at the top of every phase 1 on the wave that issued it (wave 0: +1.8k
        // cycles per tile on cfg2, the other waves waiting for it at the barrier).
        uint32_t r = tid < ng.n_rows ? tid : ng.n_rows - 1;
        asm volatile("" : "+v"(r));
        v = P.rowtab[ng.r0 - P.rowtab_row0 + r].c;
    }
    return v;
}

// scalar (wave-uniform) load of one row base
__device__ __forceinline__ RowBase load_rowbase(const ChainParams &P, uint64_t r) {
    const_f64_p rp = (const_f64_p)(uintptr_t)(P.rowtab + (r - P.rowtab_row0));
    RowBase rb;
    rb.c = rp[0]; rb.s = rp[1]; rb.nf = rp[2]; rb.pad_ = 0.0;
    return rb;
}

// row-aligned tiles (fast phase 1): base pointer of the tile's first row + a compile-time row offset -> s_load with an immediate
__device__ __forceinline__ RowBase load_rowbase_at(const_f64_p tile_rows, uint32_t i) {
    RowBase rb;
    rb.c = tile_rows[4 * i + 0]; rb.s = tile_rows[4 * i + 1]; rb.nf = tile_rows[4 * i + 2]; rb.pad_ = 0.0;
    return rb;
}

__device__ __forceinline__ uint32_t opaque_u32(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

// tile-relative sample index m -> padded LDS element  m + kPad * (m / PD)  (PD = D except for the register-tiled FIR; no pad for odd D)
template <class GeoT>
__device__ __forceinline__ uint32_t pad_index(const GeoT &geo, uint32_t m) {
    if (geo.Dp == geo.D) return m;
    if (geo.pshift != 0xffffffffu) return m + geo.kPad * (m >> geo.pshift);
    return m + geo.kPad * __umulhi(m, geo.dmagic);
}

// unpack -> NCO -> park in LDS for one fetched row.
// NCO: 0 = no shift, 1 = first-order, 2 = second-order correction.  INTERIOR rows lie wholly
// inside the tile, so no per-sample bounds checks are needed (a wave-uniform property).
template <int FMT, int NT, int NCO, bool INTERIOR, class GeoT>
__device__ __forceinline__ void process_row(const ChainParams &P, const GeoT &geo, const TileGeo &g, int32_t rel,
                                            uint32_t tid, const typename FmtTraits<FMT>::Vec &v, const RowBase &rb,
                                            const LaneRot *lr, uint32_t lane_pad, const float *lut, float2 *raw,
                                            float2 *dup = nullptr, uint32_t dup_limit = 0) {
    using FT = FmtTraits<FMT>;
    constexpr int SPL = FT::SPL;
    constexpr uint32_t ROW = NT * SPL;
    const int32_t m0 = rel + (int32_t)(tid * SPL);
    if constexpr (!INTERIOR) {
        if (!(m0 + SPL > 0 && m0 < (int32_t)g.tile_raw)) return;
    }
    float2 x[SPL];
    if constexpr (FMT == 0) {
        x[0] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
        x[1] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
    } else if constexpr (FMT == 1) {
        // arithmetic unpack: a 256-entry table in LDS costs one gathered read per component, and 64 lanes hitting a 1 KiB table
        // collide 4-5 deep in its banks — phase 1 of the cs8 chain was bound by exactly that
        const uint32_t a = v.x ^ 0x80808080u, b = v.y ^ 0x80808080u;
        x[0] = make_float2(unpack_cs8_at(a, 0), unpack_cs8_at(a, 1));
        x[1] = make_float2(unpack_cs8_at(a, 2), unpack_cs8_at(a, 3));
        x[2] = make_float2(unpack_cs8_at(b, 0), unpack_cs8_at(b, 1));
        x[3] = make_float2(unpack_cs8_at(b, 2), unpack_cs8_at(b, 3));
    } else if constexpr (FMT == 2) {
        x[0] = make_float2(unpack_cu8_at(v.x, 0), unpack_cu8_at(v.x, 1));
        x[1] = make_float2(unpack_cu8_at(v.x, 2), unpack_cu8_at(v.x, 3));
        x[2] = make_float2(unpack_cu8_at(v.y, 0), unpack_cu8_at(v.y, 1));
        x[3] = make_float2(unpack_cu8_at(v.y, 2), unpack_cu8_at(v.y, 3));
    } else {
        x[0] = make_float2(unpack_cs16(v.x & 0xffffu), unpack_cs16(v.x >> 16));
        x[1] = make_float2(unpack_cs16(v.y & 0xffffu), unpack_cs16(v.y >> 16));
        x[2] = make_float2(unpack_cs16(v.z & 0xffffu), unpack_cs16(v.z >> 16));
        x[3] = make_float2(unpack_cs16(v.w & 0xffffu), unpack_cs16(v.w >> 16));
    }
    if constexpr (NCO != 0) {
        if (!QD_DBG(P, 1)) {
            float2 m[SPL];
            nco_mul_n<NCO == 2, SPL>(rb, lr, P.ratio, m);          // the SPL f64 chains issued interleaved
#pragma unroll
            for (int u = 0; u < SPL; ++u) x[u] = cmul_pk(x[u], m[u]);   // buf[i] *= mul (src/shift.rs:51)
        }
    }
    if constexpr (GeoT::kPlanar && NT == 256) {
        // planar tile: re plane, then im plane; sample m sits at float m + (DpP - D) * (m / D) of its plane.  rel is a
        // multiple of D (tile starts and rows both are) and SPL divides D, so a lane's SPL samples are contiguous.
        constexpr uint32_t LOGD = ct_log2(GeoT::D), PADP = GeoT::DpP - GeoT::D;
        float *pre = reinterpret_cast<float *>(raw), *pim = pre + GeoT::plane_floats;
        const int32_t off = m0 + (int32_t)PADP * (m0 >> LOGD);            // arithmetic shift: floor for the (never stored) negatives
        if constexpr (INTERIOR) {
            if constexpr (SPL == 2) {
                *reinterpret_cast<float2 *>(pre + off) = make_float2(x[0].x, x[1].x);
                *reinterpret_cast<float2 *>(pim + off) = make_float2(x[0].y, x[1].y);
            } else {
                *reinterpret_cast<float4 *>(pre + off) = make_float4(x[0].x, x[1].x, x[2].x, x[3].x);
                *reinterpret_cast<float4 *>(pim + off) = make_float4(x[0].y, x[1].y, x[2].y, x[3].y);
            }
        } else {
#pragma unroll
            for (int u = 0; u < SPL; ++u)
                if (m0 + u >= 0 && m0 + u < (int32_t)g.tile_raw) { pre[off + u] = x[u].x; pim[off + u] = x[u].y; }
        }
        return;
    }
    // Additive addressing: rel is a multiple of PD (n_start and ROW both are) and SPL divides PD, so
    // pad(rel + t) = pad_s(rel) + pad(t) and a lane's SPL samples are contiguous in LDS.
    const bool additive = geo.pshift != 0xffffffffu && geo.PD >= (uint32_t)SPL && geo.PD <= ROW;
    if (QD_DBG(P, 32)) {            // timing-only ablation: consume the row without the LDS store
        asm volatile("" :: "v"(x[0].x), "v"(x[0].y), "v"(x[SPL - 1].x), "v"(x[SPL - 1].y));
        return;
    }
    if (additive) {
        const int32_t row_pad = rel + ((geo.Dp != geo.D) ? (int32_t)geo.kPad * (rel >> geo.pshift) : 0);      // uniform, signed
        float2 *dst = raw + (row_pad + (int32_t)lane_pad);
#pragma unroll
        for (int u = 0; u < SPL; ++u)
            if (INTERIOR || (m0 + u >= 0 && m0 + u < (int32_t)g.tile_raw)) dst[u] = x[u];
        if (dup != nullptr && tid * SPL < dup_limit) {        // streaming kernel: the ring's first samples once more behind its end
            float2 *d2 = dup + (row_pad + (int32_t)lane_pad);
#pragma unroll
            for (int u = 0; u < SPL; ++u) d2[u] = x[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const int32_t m = m0 + u;
            if (INTERIOR || (m >= 0 && m < (int32_t)g.tile_raw)) raw[pad_index(geo, (uint32_t)m)] = x[u];
        }
    }
}

template <int FMT, int NT, int NCO, class GeoT>
__device__ __forceinline__ void process_row_any(const ChainParams &P, const GeoT &geo, const TileGeo &g, uint32_t i,
                                                uint32_t tid, const typename FmtTraits<FMT>::Vec &v, const RowBase &rb,
                                                const LaneRot *lr, uint32_t lane_pad, const float *lut, float2 *raw) {
    constexpr uint32_t ROW = NT * FmtTraits<FMT>::SPL;
    const int32_t rel = g.rel0 + (int32_t)(i * ROW);                                              // wave-uniform
    if (rel >= 0 && rel + (int32_t)ROW <= (int32_t)g.tile_raw)
        process_row<FMT, NT, NCO, true>(P, geo, g, rel, tid, v, rb, lr, lane_pad, lut, raw);
    else
        process_row<FMT, NT, NCO, false>(P, geo, g, rel, tid, v, rb, lr, lane_pad, lut, raw);
}

// ---------------------------------------------------------------- FIR
// Taps [j0, j1) for one output; rowp points at the LDS row (of D samples, pitch Dp) holding tap j0
// at column b.  Control flow is wave-uniform; only rowp and jmax differ per lane.  With FixedGeo
// all bounds are constants and both loops unroll completely (LDS offsets become immediates).
// SNAP (shape-specialised kernels only): the truncated outputs of the reference are
// prefix sums of the same ascending-j chain, and a lane's jmax can only be T/2 + m*D.  So the wave
// runs ONE unpredicated chain over all T taps and copies the accumulator aside when the tap index
// reaches one of those compile-time-known values and equals the lane's jmax — 3 VALU ops per D
// taps instead of a predicate on every tap past T_fast.
template <bool PRED, class GeoT, bool SNAP = false>
__device__ __forceinline__ void fir_span(const GeoT &geo, const float2 *rowp, uint32_t b, uint32_t j0, uint32_t j1,
                                         uint32_t jmax, const float *__restrict__ taps, float &accr, float &acci,
                                         float2 *snap_out = nullptr) {
    const uint32_t D = geo.D, Dp = geo.Dp;
    const uint32_t n = j1 - j0;                       // taps to do
    const uint32_t n_rows = (b + n + D - 1) / D;      // LDS rows touched
    const float *h = taps + j0;                      // LDS, same address in every lane: broadcast reads
    float snapr = 0.f, snapi = 0.f;
    // is tap index jj (absolute, == j0 + relative) a possible jmax?  T/2 + m*D with m >= 1
    auto is_snap = [&](uint32_t jj) -> bool { return SNAP && jj >= geo.T / 2 + D && jj < geo.T && ((jj - geo.T / 2) % D) == 0; };
    if constexpr (GeoT::kFixed) {
        // Fully unrolled: every bound is a constant, LDS offsets are immediates.
        (void)n_rows;
        auto lds_index = [&](uint32_t jj) -> uint32_t { const uint32_t t = b + jj; return (t / D) * Dp + (t % D); };
        if (n < 64) {
            // short filters: let the scheduler interleave reads and the accumulate chain
#pragma unroll
            for (uint32_t i = 0; i < n; ++i) {
                float2 x = rowp[lds_index(i)];
                float hh = h[i];
                if (is_snap(j0 + i)) { if (jmax == j0 + i) { snapr = accr; snapi = acci; } }
                if (!PRED || (j0 + i) < jmax) {
                    accr = accr + x.x * hh;           // Complex<f32> * f32, then +=  (src/filter.rs:119)
                    acci = acci + x.y * hh;
                }
            }
        } else if (n >= 128 && GeoT::kFirBlock >= 8 && D % 8 == 0 && b % 8 == 0 && n % 8 == 0 && (!SNAP || (geo.T % D) % 8 == 0) && !PRED) {
            // long filters on 8-aligned geometry: ROLLED and software-pipelined.  8-tap blocks never straddle an LDS row
            // (D, b multiples of 8); two register sets ping-pong so that block k+1's LDS reads (8 samples
            // + two broadcast ds_read_b128 of taps) are in flight while block k is accumulated; a lane's
            // jmax (multiple of 8 here) can only be hit at a block boundary, so the snapshot is one
            // compare + two selects per candidate block.  Fully unrolling 400-500 taps instead makes
            // the register allocator spill hundreds of values.
            const uint32_t nblk = n / 8;
            float2 xa[8], xb[8]; float4 ha[2], hb[2];
            auto load_blk = [&](uint32_t blk, float2 *x, float4 *hh) {
                const uint32_t t = b + blk * 8;
                const float2 *pp = rowp + (t / D) * Dp + (t % D);          // wave-uniform offset
                if constexpr (GeoT::kPad == 2) {                           // 16-byte aligned pairs: ds_read_b128
                    const float4 *p4 = reinterpret_cast<const float4 *>(pp);
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) { const float4 v = p4[i2]; x[2 * i2] = make_float2(v.x, v.y); x[2 * i2 + 1] = make_float2(v.z, v.w); }
                } else
#pragma unroll
                for (int i2 = 0; i2 < 8; ++i2) x[i2] = pp[i2];
                const float4 *hp = reinterpret_cast<const float4 *>(h + blk * 8);
                hh[0] = hp[0]; hh[1] = hp[1];
            };
            auto mac_blk = [&](uint32_t blk, const float2 *x, const float4 *hh) {
                const uint32_t jj = j0 + blk * 8;
                if (SNAP && jj >= geo.T / 2 + D && jj < geo.T && ((jj - geo.T / 2) % D) == 0) {   // wave-uniform
                    if (jmax == jj) { snapr = accr; snapi = acci; }
                }
                if constexpr ((GeoT::kFlags & kGeoPackedSpan) != 0) {
                    // both chains of the output as one float pair: v_pk_mul_f32 + v_pk_add_f32 per tap (same two roundings per
                    // component; half the VALU instructions — the kernel is bound by issue slots, see fir_pair)
                    typedef float v2f_t __attribute__((ext_vector_type(2)));
                    v2f_t a2 = {accr, acci};
                    const float hv[8] = {hh[0].x, hh[0].y, hh[0].z, hh[0].w, hh[1].x, hh[1].y, hh[1].z, hh[1].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const v2f_t xv = {x[i].x, x[i].y}; const v2f_t hs = {hv[i], hv[i]}; a2 = a2 + xv * hs; }
                    accr = a2.x; acci = a2.y;
                    return;
                }
                accr = accr + x[0].x * hh[0].x; acci = acci + x[0].y * hh[0].x;
                accr = accr + x[1].x * hh[0].y; acci = acci + x[1].y * hh[0].y;
                accr = accr + x[2].x * hh[0].z; acci = acci + x[2].y * hh[0].z;
                accr = accr + x[3].x * hh[0].w; acci = acci + x[3].y * hh[0].w;
                accr = accr + x[4].x * hh[1].x; acci = acci + x[4].y * hh[1].x;
                accr = accr + x[5].x * hh[1].y; acci = acci + x[5].y * hh[1].y;
                accr = accr + x[6].x * hh[1].z; acci = acci + x[6].y * hh[1].z;
                accr = accr + x[7].x * hh[1].w; acci = acci + x[7].y * hh[1].w;
            };
            // branch-free body in pairs (the scheduler is kept from hoisting set A's reload above its last use, which
            // would cost a register copy of the whole set at the back edge); an odd last block is peeled
            load_blk(0, xa, ha);
            uint32_t blk = 0;
#pragma unroll 1
            for (; blk + 2 < nblk; blk += 2) {      // never fully unrolled: at T = 800 that spills ~1900 VGPRs
                load_blk(blk + 1, xb, hb);
                mac_blk(blk, xa, ha);
                __builtin_amdgcn_sched_barrier(0);
                load_blk(blk + 2, xa, ha);
                mac_blk(blk + 1, xb, hb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (blk + 1 < nblk) { load_blk(blk + 1, xb, hb); mac_blk(blk, xa, ha); mac_blk(blk + 1, xb, hb); }
            else mac_blk(blk, xa, ha);
        } else if (n > 256 && GeoT::kFirBlock >= 8) {
            // very long filters: a ROLLED loop over LDS rows (D taps each) with an 8-tap inner unroll —
            // fully unrolling 400-500 taps blows the register allocator up.  A lane's jmax can only be
            // reached at ONE column of a row (column T % D, see is_snap), so the snapshot check is
            // peeled to that column: one compare + two selects per row instead of per tap.
            const uint32_t bstar = geo.T % D;                  // column where (tap - T/2) % D == 0
            const float2 *rp = rowp;
            uint32_t j = 0, bcur = b;
            while (j < n) {                                    // wave-uniform trip structure
                uint32_t run = D - bcur;
                if (run > n - j) run = n - j;
                // split [bcur, bcur+run) at bstar
                uint32_t first = run;
                if (SNAP && bstar >= bcur && bstar < bcur + run) first = bstar - bcur;
#pragma unroll 8
                for (uint32_t i = 0; i < first; ++i) {
                    float2 x = rp[bcur + i]; float hh = h[j + i];
                    if (!PRED || (j0 + j + i) < jmax) { accr = accr + x.x * hh; acci = acci + x.y * hh; }
                }
                if (SNAP && first < run) {
                    if (is_snap(j0 + j + first)) { if (jmax == j0 + j + first) { snapr = accr; snapi = acci; } }
#pragma unroll 8
                    for (uint32_t i = first; i < run; ++i) {
                        float2 x = rp[bcur + i]; float hh = h[j + i];
                        if (!PRED || (j0 + j + i) < jmax) { accr = accr + x.x * hh; acci = acci + x.y * hh; }
                    }
                }
                j += run; bcur = 0; rp += Dp;
            }
        } else {
            // long filters, software-pipelined by hand: the LDS reads of block k+1 (8 taps: samples +
            // tap values) are issued before block k is consumed, so the dependent multiply/accumulate
            // chain does not stall on LDS latency after every pair.
            constexpr uint32_t B = GeoT::kFirBlock;
            float2 xa[B]; float ha[B];
#pragma unroll
            for (uint32_t i = 0; i < B; ++i) if (i < n) { xa[i] = rowp[lds_index(i)]; ha[i] = h[i]; }
#pragma unroll
            for (uint32_t base = 0; base < n; base += B) {
                float2 xb[B]; float hb[B];
#pragma unroll
                for (uint32_t i = 0; i < B; ++i) if (base + B + i < n) { xb[i] = rowp[lds_index(base + B + i)]; hb[i] = h[base + B + i]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t i = 0; i < B; ++i) {
                    if (base + i < n) {
                        if (is_snap(j0 + base + i)) { if (jmax == j0 + base + i) { snapr = accr; snapi = acci; } }
                        if (!PRED || (j0 + base + i) < jmax) {
                            accr = accr + xa[i].x * ha[i];
                            acci = acci + xa[i].y * ha[i];
                        }
                    }
                }
                asm volatile("" : "+v"(accr), "+v"(acci));     // pin the add chain inside its block (see fir_comp)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t i = 0; i < B; ++i) { xa[i] = xb[i]; ha[i] = hb[i]; }
            }
        }
        if (SNAP) {
            if (snap_out) *snap_out = make_float2(snapr, snapi);        // caller wants both the full and the truncated value
            else if (jmax < geo.T) { accr = snapr; acci = snapi; }
        }
    } else {
        uint32_t j = 0;
        while (j < n) {
            uint32_t run = D - b;
            if (run > n - j) run = n - j;
            const float2 *p = rowp + b;
#pragma unroll 8
            for (uint32_t i = 0; i < run; ++i) {
                float2 x = p[i];
                float hh = h[j + i];
                if (!PRED || (j0 + j + i) < jmax) {
                    accr = accr + x.x * hh;
                    acci = acci + x.y * hh;
                }
            }
            j += run;
            b = 0;
            rowp += Dp;
        }
    }
}

// Register-tiled FIR (shape-specialised kernels, long filters): one lane computes R consecutive
// outputs q0..q0+R-1.  The lane walks the L = (R-1)*D + T samples its outputs touch ONCE (BS-sample
// blocks, ping-ponged registers); sample i feeds accumulator r with tap i - r*D.  Every accumulator
// still sees its products in ascending-j order with separately rounded multiply and add, so the
// results are bit-identical to the one-output-per-lane form, but LDS sample traffic drops R-fold
// (the one-output form reads 8 B of LDS per MAC and is LDS-bandwidth-bound for T/D >> 1).
// Truncated outputs (jmax[r] < T) are accumulator snapshots, as in fir_span.
// BS = 4 keeps the two register sets (samples + R tap vectors each) at 16 + 8R VGPRs: the phase-1
// prefetch registers stay live across the FIR, so the budget here is ~90 of the 128.
template <int R, class GeoT, int BS = 4>
__device__ __forceinline__ void fir_tiled(const float2 *lanep, const uint32_t *jmax, const float *h, float2 *full, float2 *snap) {
    constexpr uint32_t D = GeoT::D, T = GeoT::T, c = GeoT::c;
    constexpr uint32_t L = (R - 1) * D + T, NB = L / BS;
    static_assert(GeoT::PD == R * D && GeoT::pshift != 0xffffffffu, "pad period of the register-tiled layout");
    static_assert(BS == 4 || BS == 8, "block of 4 or 8 samples");
    static_assert(L % 8 == 0 && D % 8 == 0 && c % 8 == 0, "register-tiled FIR needs 8-aligned geometry");
    float ar[R], ai[R], sr[R], si[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ar[r] = 0.f; ai[r] = 0.f; sr[r] = 0.f; si[r] = 0.f; }
    auto load_x = [&](uint32_t blk, float2 *x) {
        // lanep is the padded LDS address of sample q0*D (a multiple of the pad period), so pad(q0*D + t) =
        // pad(q0*D) + pad(t) with a wave-uniform second term; a block never straddles a pad
        const uint32_t t = c + blk * BS;
        const float2 *pp = lanep + (t + GeoT::kPad * (t >> GeoT::pshift));
        if constexpr (GeoT::kPad == 2) {                              // 16-byte aligned pairs: ds_read_b128
            const float4 *p4 = reinterpret_cast<const float4 *>(pp);
#pragma unroll
            for (int i = 0; i < BS / 2; ++i) { const float4 v = p4[i]; x[2 * i] = make_float2(v.x, v.y); x[2 * i + 1] = make_float2(v.z, v.w); }
        } else {
            const uint64_t *p8 = reinterpret_cast<const uint64_t *>(pp);  // one 8-byte LDS read per sample (rows are 8-byte aligned only)
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                const uint64_t v = p8[i];
                x[i] = make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
            }
        }
    };
    auto load_h1 = [&](uint32_t j0, float *hh) {                   // BS taps from h + j0 (16-byte aligned broadcast reads)
        const float4 *hp = reinterpret_cast<const float4 *>(h + j0);
#pragma unroll
        for (int v = 0; v < BS / 4; ++v) {
            const float4 q = hp[v];
            hh[4 * v + 0] = q.x; hh[4 * v + 1] = q.y; hh[4 * v + 2] = q.z; hh[4 * v + 3] = q.w;
        }
    };
    auto load_h = [&](uint32_t blk, float (*hh)[BS]) {
#pragma unroll
        for (int r = 0; r < R; ++r) load_h1(blk * BS - r * D, hh[r]);
    };
    // only a wave that owns truncated outputs pays for the snapshot compares
    bool lane_trunc = false;
#pragma unroll
    for (int r = 0; r < R; ++r) lane_trunc |= jmax[r] < T;
    const bool need_snap = __builtin_amdgcn_ballot_w64(lane_trunc) != 0;
    auto snap_check = [&](int r, uint32_t j0) {                     // j0 wave-uniform
        if (need_snap && j0 >= T / 2 + D && ((j0 - T / 2) % D) == 0) {
            if (jmax[r] == j0) { sr[r] = ar[r]; si[r] = ai[r]; }
        }
    };
    // interior block (every accumulator active): the R chains are interleaved tap by tap so that
    // consecutive VALU ops are independent; its taps were fetched one block ahead with the samples
    auto mac_int = [&](uint32_t blk, const float2 *x, const float (*hh)[BS]) {
#pragma unroll
        for (int r = 0; r < R; ++r) snap_check(r, blk * BS - r * D);
#pragma unroll
        for (int i = 0; i < BS; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ar[r] = ar[r] + x[i].x * hh[r][i];
                ai[r] = ai[r] + x[i].y * hh[r][i];
            }
        }
    };
    // edge block (compile-time blk): only the accumulators whose tap range covers it take part
    auto mac_edge = [&](uint32_t blk, const float2 *x) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int32_t j0 = (int32_t)(blk * BS) - r * (int32_t)D;
            if (j0 >= 0 && j0 < (int32_t)T) {
                snap_check(r, (uint32_t)j0);
                float hh[BS];
                load_h1((uint32_t)j0, hh);
#pragma unroll
                for (int i = 0; i < BS; ++i) {
                    ar[r] = ar[r] + x[i].x * hh[i];
                    ai[r] = ai[r] + x[i].y * hh[i];
                }
            }
        }
    };
    constexpr uint32_t P = (R - 1) * D / BS, E = T / BS;        // interior blocks are [P, E)
    static_assert(E > P + 2, "filter too short for the register-tiled FIR");
    constexpr uint32_t P2 = P + ((E - P) & 1);                  // even interior trip count
    float2 xa[BS], xb[BS];
    float ha[R][BS], hb[R][BS];
#pragma unroll
    for (uint32_t blk = 0; blk < P2; ++blk) {
        load_x(blk, xa);
        if (blk < P) mac_edge(blk, xa); else { load_h(blk, ha); mac_int(blk, xa, ha); }
    }
    load_x(P2, xa); load_h(P2, ha);
    for (uint32_t blk = P2; blk + 2 < E; blk += 2) {              // branch-free body: two register sets, no copies
        load_x(blk + 1, xb); load_h(blk + 1, hb);
        mac_int(blk, xa, ha);
        __builtin_amdgcn_sched_barrier(0);     // keep set A's reload below its last use: no register copies at the back edge
        load_x(blk + 2, xa); load_h(blk + 2, ha);
        mac_int(blk + 1, xb, hb);
        __builtin_amdgcn_sched_barrier(0);
    }
    load_x(E - 1, xb); load_h(E - 1, hb);                         // last interior pair, peeled
    mac_int(E - 2, xa, ha);
    load_x(E, xa);                                                // the first edge block (always inside the tile)
    mac_int(E - 1, xb, hb);
#pragma unroll
    for (uint32_t blk = E; blk < NB; ++blk) {
        if (blk > E) load_x(blk, xa);
        mac_edge(blk, xa);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { full[r] = make_float2(ar[r], ai[r]); snap[r] = make_float2(sr[r], si[r]); }
}

// Component-split FIR (shape-specialised kernels whose tile has at most NT/2 outputs): a lane carries ONE of
// the two independent accumulate chains of an output — acc.re += x.re*h or acc.im += x.im*h, exactly the
// reference's two f32 chains (src/filter.rs:119) — so a tile with few outputs still fills the workgroup and
// each lane reads 4 B of LDS per tap.  Same products, same order; only the lane that owns them changes.
// xp points at this component of the first sample of the output's first LDS row.  8-tap blocks, two register
// sets (block k+1's reads in flight while block k accumulates); a truncated output (jmax < T) is the
// accumulator snapshot taken at tap jmax, as in fir_span.
template <class GeoT>
__device__ __forceinline__ float fir_comp(const float *xp, uint32_t jmax, const float *h, float *snap_out = nullptr) {
    constexpr uint32_t D = GeoT::D, Dp = GeoT::Dp, T = GeoT::T, b = GeoT::b0;
    // PRE taps reach the first 8-aligned LDS column; from there 8-tap blocks never straddle a row (D % 8 == 0),
    // so the body is a ROLLED loop (fully unrolled, the scheduler hoists every read and spills ~150 VGPRs)
    constexpr uint32_t PRE = (8 - b % 8) % 8, NBLK = (T - PRE) / 8, TAIL = (T - PRE) % 8;
    constexpr uint32_t CPOS = (T / 2 + 8 * D - PRE) % 8;          // where in a block a possible jmax (T/2 + m*D) can sit
    static_assert(D % 8 == 0 && NBLK >= 3, "component-split FIR geometry");
    float acc = 0.f, snap = 0.f;
    auto cand = [&](uint32_t jj) -> bool { return jj >= T / 2 + D && jj < T && ((jj - T / 2) % D) == 0; };
    auto xoff = [&](uint32_t t) -> uint32_t { return 2 * ((t / D) * Dp + (t % D)); };
#pragma unroll
    for (uint32_t i = 0; i < PRE; ++i) {                                   // head (compile-time offsets)
        if (cand(i)) { if (jmax == i) snap = acc; }
        acc = acc + xp[xoff(b + i)] * h[i];
    }
    float xa[8], xb[8], ha[8], hb[8];
    auto load = [&](uint32_t m, float *x, float *hh) {
        const uint32_t t = b + PRE + 8 * m;                                // wave-uniform, multiple of 8
        const float *pp = xp + xoff(t);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = pp[2 * i];
        if constexpr (PRE % 4 == 0) {
            const float4 *hp = reinterpret_cast<const float4 *>(h + PRE + 8 * m);
            const float4 h0 = hp[0], h1 = hp[1];
            hh[0] = h0.x; hh[1] = h0.y; hh[2] = h0.z; hh[3] = h0.w; hh[4] = h1.x; hh[5] = h1.y; hh[6] = h1.z; hh[7] = h1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hh[i] = h[PRE + 8 * m + i];
        }
    };
    auto mac = [&](uint32_t m, const float *x, const float *hh) {
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            if (i == CPOS) { const uint32_t jj = PRE + 8 * m + i; if (cand(jj) && jmax == jj) snap = acc; }
            acc = acc + x[i] * hh[i];
        }
    };
    constexpr uint32_t NPAIR = NBLK / 2;                                   // blocks [0, 2*NPAIR) in ping-pong pairs
    load(0, xa, ha);
    // unroll explicitly: hipcc fully unrolls this loop for T = 200 on its own, hiprtc does not (10 % slower kernel)
    // The accumulator is pinned at the end of every block (an empty asm that "modifies" it): sched_barrier only orders
    // instructions with side effects, so without the pin instruction selection is free to sink the whole dependent add
    // chain below the last sched_barrier of a fully unrolled body — every product is then computed early and SPILLED
    // (185 VGPRs of scratch and 8.7x the run time on the 200-tap shape when a never-taken branch in front of this call
    // was removed).
    auto pair = [&](uint32_t m) {
        load(m + 1, xb, hb);
        mac(m, xa, ha);
        asm volatile("" : "+v"(acc));
        __builtin_amdgcn_sched_barrier(0);      // set A is reloaded only below its last use: no copies at the back edge
        load(m + 2, xa, ha);
        mac(m + 1, xb, hb);
        asm volatile("" : "+v"(acc));
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NPAIR <= 16) {
#pragma unroll
        for (uint32_t m = 0; m + 2 < 2 * NPAIR; m += 2) pair(m);
    } else {
#pragma unroll 1
        for (uint32_t m = 0; m + 2 < 2 * NPAIR; m += 2) pair(m);      // longer filters stay rolled (a full unroll of T = 800 spills ~1900 VGPRs;
                                                                       // `#pragma unroll 2` is unrolled again, fully, by a later pass)
    }
    load(2 * NPAIR - 1, xb, hb);
    mac(2 * NPAIR - 2, xa, ha);
    if constexpr (NBLK % 2) load(NBLK - 1, xa, ha);
    mac(2 * NPAIR - 1, xb, hb);
    if constexpr (NBLK % 2) mac(NBLK - 1, xa, ha);
#pragma unroll
    for (uint32_t i = 0; i < TAIL; ++i) {                                  // tail
        const uint32_t jj = PRE + 8 * NBLK + i;
        if (cand(jj)) { if (jmax == jj) snap = acc; }
        acc = acc + xp[xoff(b + jj)] * h[jj];
    }
    if (snap_out) { *snap_out = snap; return acc; }      // shared-FIR mode keeps both (dec / trc)
    return jmax < T ? snap : acc;
}

// Component-split FIR over the PLANAR tile (FixedGeo::kPlanar): a lane carries one accumulate chain — acc.re += x.re*h or
// acc.im += x.im*h, the reference's two f32 chains (src/filter.rs:119), ascending taps, separately rounded multiply and add —
// and reads FOUR consecutive taps' worth of its component with one ds_read_b128 (the plane's rows are 16-byte aligned at a
// pitch whose quarter is odd: conflict-free in every 16-lane group).  xp = this lane's plane + the first float of the output's
// first LDS row.  PF blocks are kept in flight.  With baked taps (plan-time builds) the taps are immediates; otherwise they are
// broadcast ds_read_b128 from LDS.  A truncated output (jmax < T) is the accumulator snapshot at tap jmax, as in fir_span.
template <class GeoT>
__device__ __forceinline__ float fir_comp_planar(const float *xp, uint32_t jmax, const float *h) {
    constexpr uint32_t D = GeoT::D, DpP = GeoT::DpP, T = GeoT::T, b = GeoT::b0, NB = T / 4;
    constexpr int PF = GeoT::kFirBlock >= 4 ? (int)(GeoT::kFirBlock / 2) : 2;     // blocks of 4 taps in flight (FIRB 8 -> 4 blocks = 16 taps ahead)
    static_assert(T % 4 == 0 && b % 4 == 0 && D % 4 == 0 && NB > (uint32_t)PF, "planar FIR geometry");
    float acc = 0.f, snap = 0.f;
    auto cand = [&](uint32_t jj) -> bool { return jj >= T / 2 + D && jj < T && ((jj - T / 2) % D) == 0; };
    auto xoff = [&](uint32_t t) -> uint32_t { return (t / D) * DpP + (t % D); };
    float4 x[PF], hh[PF];
    auto load = [&](uint32_t k, int slot) {
        x[slot] = *reinterpret_cast<const float4 *>(xp + xoff(b + 4 * k));
        if constexpr (!GeoT::kBakedTaps) hh[slot] = *reinterpret_cast<const float4 *>(h + 4 * k);
    };
    auto tap = [&](uint32_t jj, int slot, int i) -> float {
        if constexpr (GeoT::kBakedTaps) return kBakedTapTable[jj < sizeof(kBakedTapTable) / sizeof(float) ? jj : 0];
        else return i == 0 ? hh[slot].x : (i == 1 ? hh[slot].y : (i == 2 ? hh[slot].z : hh[slot].w));
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load(k, k);
#pragma unroll
    for (uint32_t k = 0; k < NB; ++k) {
        const int slot = (int)(k % PF);
        const float xs[4] = {x[slot].x, x[slot].y, x[slot].z, x[slot].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t jj = 4 * k + i;
            if (cand(jj)) { if (jmax == jj) snap = acc; }
            acc = acc + xs[i] * tap(jj, slot, i);
        }
        asm volatile("" : "+v"(acc));                 // pin the add chain inside its block (see fir_comp)
        if (k + PF < NB) load(k + PF, slot);           // refill the slot just consumed
        __builtin_amdgcn_sched_barrier(0);
    }
    return jmax < T ? snap : acc;
}

// Packed lane-per-output FIR (FixedGeo FLAGS_ bit 2, interleaved tile with 16-byte aligned rows, PAD 2): a lane carries BOTH
// accumulate chains of one output as a float pair and advances them with one v_pk_mul_f32 + one v_pk_add_f32 per tap —
// the same two separately rounded operations per component as the scalar form (src/filter.rs:119), half the VALU
// instructions.  On gfx950 a packed op occupies the pipe as long as two scalar ones, but the chain kernel is bound by
// issue slots, not by pipe time: the component-split FIR issues 4 VALU instructions per tap and output, this one 2.  Only
// G*W lanes take part (half the workgroup at G*W = 128).  Two taps (re, im, re, im) come with each ds_read_b128.
template <class GeoT, bool PACKED = true>
__device__ __forceinline__ float2 fir_pair(const float2 *rowp /* first LDS row of the output */, uint32_t jmax, const float *h, float2 *snap_out = nullptr) {
    constexpr uint32_t D = GeoT::D, Dp = GeoT::Dp, T = GeoT::T, b = GeoT::b0, NB = T / 4;     // blocks of 4 taps = two b128 sample reads
    constexpr int PF = 3;
    static_assert(GeoT::kPad == 2 && T % 4 == 0 && b % 2 == 0 && D % 4 == 0 && NB > (uint32_t)PF, "packed FIR geometry");
#if defined(QD_FIR_ABL) && (QD_FIR_ABL & 4)
    if (snap_out) *snap_out = rowp[1];                     // timing-only ablation (development builds): no FIR at all
    return rowp[0];
#endif
    v2f acc = {0.f, 0.f}, snap = {0.f, 0.f};
    float ar = 0.f, ai = 0.f, sr = 0.f, si = 0.f;          // the scalar form keeps plain floats (a vector type would be re-packed)
    auto cand = [&](uint32_t jj) -> bool { return jj >= T / 2 + D && jj < T && ((jj - T / 2) % D) == 0; };
    auto xoff = [&](uint32_t t) -> uint32_t { return (t / D) * Dp + (t % D); };      // float2 elements
    float4 xa[PF], xb[PF], hh[PF];
    auto load = [&](uint32_t k, int slot) {
        const uint32_t t0 = b + 4 * k;                      // taps 4k..4k+3: sample pairs (4k, 4k+1) and (4k+2, 4k+3) never straddle a row (D % 4 == 0, b even)
        xa[slot] = *reinterpret_cast<const float4 *>(rowp + xoff(t0));
        xb[slot] = *reinterpret_cast<const float4 *>(rowp + xoff(t0 + 2));
        if constexpr (!GeoT::baked_request) hh[slot] = *reinterpret_cast<const float4 *>(h + 4 * k);
    };
    auto tap = [&](uint32_t jj, int slot, int i) -> float {
        if constexpr (GeoT::baked_request) return kBakedTapTable[jj < sizeof(kBakedTapTable) / sizeof(float) ? jj : 0];
        else return i == 0 ? hh[slot].x : (i == 1 ? hh[slot].y : (i == 2 ? hh[slot].z : hh[slot].w));
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load(k, k);
    if constexpr (PACKED && !GeoT::baked_request) {
        // Taps from LDS, four per ds_read_b128, used straight out of the register pair they arrive in: op_sel picks the pair's
        // low or high word for BOTH halves of the packed multiply, so no tap is ever splat into a register pair of its own
        // (hipcc emits a v_mov per tap for that, or an s_mov per tap for immediates).  One asm statement per block of four taps:
        // 4 v_pk_mul_f32 + 4 v_pk_add_f32 in chain order — the statement is also what keeps the adds next to their products.
        // 11 instructions per four taps (3 LDS reads + 8 VALU) against 18 with the taps as immediates.
#pragma unroll
        for (uint32_t k = 0; k < NB; ++k) {
            const int slot = (int)(k % PF);
            if (cand(4 * k)) { if (jmax == 4 * k) snap = acc; }                 // candidates are multiples of four (D % 4 == 0, T/2 % 4 == 0)
            static_assert((T / 2) % 4 == 0, "snapshot taps sit on block boundaries");
            const v2f x0 = {xa[slot].x, xa[slot].y}, x1 = {xa[slot].z, xa[slot].w}, x2 = {xb[slot].x, xb[slot].y}, x3 = {xb[slot].z, xb[slot].w};
            const v2f h01 = {hh[slot].x, hh[slot].y}, h23 = {hh[slot].z, hh[slot].w};
            v2f p0, p1;
            if constexpr ((GeoT::kFlags & kGeoFastFma) != 0) {
                // QD_MODE_FAST: acc = fma(x, h, acc) — the same ascending-tap chain with ONE rounding per tap; not the reference's bits
                asm volatile("v_pk_fma_f32 %0, %1, %5, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                             "v_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                             : "+v"(acc)
                             : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
            } else
            asm volatile("v_pk_mul_f32 %1, %3, %7 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %2, %4, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %1\n\t"
                         "v_pk_mul_f32 %1, %5, %8 op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %0, %0, %2\n\t"
                         "v_pk_mul_f32 %2, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %1\n\t"
                         "v_pk_add_f32 %0, %0, %2"
                         : "+v"(acc), "=&v"(p0), "=&v"(p1)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
            if (k + PF < NB) load(k + PF, slot);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (snap_out) { *snap_out = make_float2(snap.x, snap.y); return make_float2(acc.x, acc.y); }
        const v2f r = jmax < T ? snap : acc;
        return make_float2(r.x, r.y);
    }
#pragma unroll
    for (uint32_t k = 0; k < NB; ++k) {
        const int slot = (int)(k % PF);
        const float xr[4] = {xa[slot].x, xa[slot].z, xb[slot].x, xb[slot].z}, xi[4] = {xa[slot].y, xa[slot].w, xb[slot].y, xb[slot].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t jj = 4 * k + i;
            const float hv = tap(jj, slot, i);
            if constexpr (PACKED) {
                if (cand(jj)) { if (jmax == jj) snap = acc; }
                const v2f xs = {xr[i], xi[i]}, hs = {hv, hv};
                acc = acc + xs * hs;                       // (re, im) * h, then +=: one v_pk_mul_f32, one v_pk_add_f32
            } else {                                       // two scalar chains (FIR-dominated shapes: shorter dependent latency per tap)
                if (cand(jj)) { if (jmax == jj) { sr = ar; si = ai; } }
                ar = ar + xr[i] * hv;
                ai = ai + xi[i] * hv;
            }
        }
        if constexpr (PACKED) asm volatile("" : "+v"(acc));      // pin the add chain(s) inside their block (see fir_comp)
        else asm volatile("" : "+v"(ar), "+v"(ai));
        if (k + PF < NB) load(k + PF, slot);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!PACKED) { acc.x = ar; acc.y = ai; snap.x = sr; snap.y = si; }
    if (snap_out) { *snap_out = make_float2(snap.x, snap.y); return make_float2(acc.x, acc.y); }      // shared-FIR mode keeps both
    const v2f r = jmax < T ? snap : acc;
    return make_float2(r.x, r.y);
}

// Register-tiled FIR, two outputs per lane, as straight-line packed code (helper-mode kernels: no truncated outputs on the main
// lanes).  Same walk as fir_tiled<2>: the lane reads the D + T samples its two outputs touch once, sample i feeds output 0 with
// tap i and output 1 with tap i - D, every chain in ascending-tap order with separately rounded multiply and add.  What
// changes is the instruction stream: fir_tiled's rolled loop spends 75 instructions per 8 taps (32 packed VALU, 8 register
// copies to splat taps, ~20 scalar address operations, LDS reads through freshly materialised address registers) and a wave
// issues one instruction per ~5 cycles — with two FIR waves per SIMD the phase is bound by instruction issue, not by the
// VALU.  Here every LDS offset is an immediate, a tap is used straight out of the register pair it arrives in (op_sel picks
// the word for both halves of the packed multiply), and the tap blocks are shared between the two outputs (output 1 uses the
// block output 0 used D/4 blocks earlier): 20 instructions per 4 samples (16 VALU, 3 LDS reads, 1 wait).
template <class GeoT, int TURN>
__device__ __forceinline__ void fir_tiled2_pk(const float2 *lanep, const float *h, float2 *full, const uint32_t *jm, float2 *snap) {
    constexpr uint32_t D = GeoT::D, T = GeoT::T, c = GeoT::c;
    constexpr uint32_t L = D + T, NB = L / 4, HB = T / 4, LAG = D / 4;      // sample blocks, tap blocks, output 1's lag in blocks
    // Output 1 uses the tap block output 0 used LAG blocks earlier.  Short lags (D <= 16) keep those blocks live in registers;
    // longer ones (D = 32: 8 blocks = 32 VGPRs) read the block a second time instead — a broadcast ds_read_b128, one LDS
    // bank row — so the FIR fits a 128-VGPR kernel beside the phase-1 prefetch registers.
    constexpr bool kReTap = LAG > 4;
    constexpr int PF = 3, NS = kReTap ? PF : PF + (int)LAG + 1;              // sample / tap blocks in flight, live tap slots

    static_assert(GeoT::kFirTile == 2 && GeoT::PD == 2 * D && GeoT::pshift != 0xffffffffu && GeoT::kPad == 2, "two-output packed FIR: layout");
    static_assert(D % 4 == 0 && T % 4 == 0 && c % 4 == 0 && (T / 2) % 4 == 0 && NB > (uint32_t)PF && LAG >= 1 && LAG <= 8, "two-output packed FIR: geometry");
    v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
    // Truncated outputs (the last kNtrunc of a window) are prefixes of their own full chain: the accumulator is
    // copied when the chain reaches the lane's jmax.  Only the wave that owns such outputs takes the (wave-uniform) branches.
    const bool need_snap = __builtin_amdgcn_ballot_w64(jm[0] < T || jm[1] < T) != 0;
    auto cand = [&](uint32_t j) -> bool { return j >= T / 2 + D && j < T && ((j - T / 2) % D) == 0; };
    float4 xa[PF], xb[PF], hh[NS], gg[kReTap ? PF : 1];
    // LDS address of the taps, kept in ONE vector register: every tap read is base + immediate (left uniform, hipcc forms one
    // scalar address per block — 128 scalars, spilled to vector lanes — and moves each into a vector register for its read)
    typedef float f4n __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const f4n lds_f4;
    uint32_t hbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)h;
    asm volatile("" : "+v"(hbase));
    const lds_f4 *hl = (const lds_f4 *)(uintptr_t)hbase;
    auto load = [&](uint32_t b) {
#if defined(QD_FIR_ABL) && (QD_FIR_ABL & 2)
        if (b >= (uint32_t)PF) return;                                       // timing-only ablation (development builds): no LDS reads past the first blocks
#endif
        const uint32_t t = c + 4 * b;                                        // a block of four never straddles a pad (pad period 2D, a multiple of 4)
        const float2 *pp = lanep + (t + GeoT::kPad * (t >> GeoT::pshift));
        xa[b % PF] = *reinterpret_cast<const float4 *>(pp);
        xb[b % PF] = *reinterpret_cast<const float4 *>(pp + 2);
        if (b < HB) { const f4n q = hl[b]; hh[b % NS] = make_float4(q.x, q.y, q.z, q.w); }
        if constexpr (kReTap) { if (b >= LAG) { const f4n q = hl[b - LAG]; gg[b % PF] = make_float4(q.x, q.y, q.z, q.w); } }
    };
#pragma unroll
    for (int b = 0; b < PF; ++b) load(b);
#pragma clang loop unroll(full)
    for (uint32_t b = 0; b < NB; ++b) {
        const float4 A = xa[b % PF], B = xb[b % PF];
        const v2f x0 = {A.x, A.y}, x1 = {A.z, A.w}, x2 = {B.x, B.y}, x3 = {B.z, B.w};
        v2f t0, t1, t2, t3;
        // Two FIR waves share a SIMD and the issue arbiter serves the older one first: left alone, one wave runs at full speed,
        // finishes at 60 % of the phase and leaves the other to run the tail on its own — a single wave cannot keep the VALU busy
        // (2.2 against 1.77 ns per packed op, LDS waits exposed).  The two take turns at the higher priority, 8 blocks at a time,
        // so both stay in flight to the end of the phase.
        if constexpr (TURN >= 0) { if ((b % 8) == 0) { if (((b / 8) & 1u) != (uint32_t)TURN) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); } }
        if ((b < HB && cand(4 * b)) || (b >= LAG && cand(4 * (b - LAG)))) {  // compile-time: a snapshot tap starts at this block
            if (need_snap) {
                if (b < HB && cand(4 * b)) { if (jm[0] == 4 * b) s0 = a0; }
                if (b >= LAG && cand(4 * (b - LAG))) { if (jm[1] == 4 * (b - LAG)) s1 = a1; }
            }
        }
#if defined(QD_FIR_ABL) && (QD_FIR_ABL & 1)
        {                                                                    // timing-only ablation (development builds): LDS reads without the arithmetic
            asm volatile("" :: "v"(x0), "v"(x3), "v"(hh[b % NS].x));
            if (b + PF < NB) load(b + PF);
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
#endif
        if (b >= LAG && b < HB) {                                            // both outputs take this block
            const float4 H = hh[b % NS], G = kReTap ? gg[b % PF] : hh[(b - LAG) % NS];
            const v2f h01 = {H.x, H.y}, h23 = {H.z, H.w}, g01 = {G.x, G.y}, g23 = {G.z, G.w};
            if constexpr ((GeoT::kFlags & kGeoFastFma) != 0) {                // QD_MODE_FAST: fused, two interleaved chains
                asm volatile("v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %1, %2, %8, %1 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                             "v_pk_fma_f32 %1, %3, %8, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                             "v_pk_fma_f32 %0, %4, %7, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %1, %4, %9, %1 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %5, %7, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                             "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                             : "+v"(a0), "+v"(a1)
                             : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23), "v"(g01), "v"(g23));
            } else
            asm volatile("v_pk_mul_f32 %2, %6, %10 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %3, %6, %12 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %4, %7, %10 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %2\n\t"
                         "v_pk_mul_f32 %5, %7, %12 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %1, %1, %3\n\t"
                         "v_pk_mul_f32 %2, %8, %11 op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %0, %0, %4\n\t"
                         "v_pk_mul_f32 %3, %8, %13 op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %1, %1, %5\n\t"
                         "v_pk_mul_f32 %4, %9, %11 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %2\n\t"
                         "v_pk_mul_f32 %5, %9, %13 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %1, %1, %3\n\t"
                         "v_pk_add_f32 %0, %0, %4\n\t"
                         "v_pk_add_f32 %1, %1, %5"
                         : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23), "v"(g01), "v"(g23));
        } else {                                                             // the first LAG blocks: output 0 only; the last LAG: output 1 only
            const float4 H = b < HB ? hh[b % NS] : (kReTap ? gg[b % PF] : hh[(b - LAG) % NS]);
            const v2f h01 = {H.x, H.y}, h23 = {H.z, H.w};
            v2f &acc = b < HB ? a0 : a1;
            if constexpr ((GeoT::kFlags & kGeoFastFma) != 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %5, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                             "v_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                             : "+v"(acc)
                             : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
            } else
            asm volatile("v_pk_mul_f32 %1, %3, %7 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %2, %4, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %1\n\t"
                         "v_pk_mul_f32 %1, %5, %8 op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %0, %0, %2\n\t"
                         "v_pk_mul_f32 %2, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %0, %1\n\t"
                         "v_pk_add_f32 %0, %0, %2"
                         : "+v"(acc), "=&v"(t0), "=&v"(t1)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
        }
        if (b + PF < NB) load(b + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
    full[0] = make_float2(a0.x, a0.y);
    full[1] = make_float2(a1.x, a1.y);
    snap[0] = make_float2(s0.x, s0.y);
    snap[1] = make_float2(s1.x, s1.y);
}

// Truncated tail outputs on spare lanes (register-tiled kernels whose workgroup has idle waves in the FIR phase):
// the reference's per-read_at truncation makes the last kNtrunc outputs of a window PREFIXES of the full chain
// (jmax = T/2 + m*D taps).  Computing them as accumulator snapshots inside the main loop makes the one wave that
// owns them the straggler of the phase (cfg4); here an otherwise idle wave computes each prefix directly, ascending
// taps, same products — and the main waves carry no snapshot logic at all.  (The helper's SIMD still ends ~15 %
// after the others: two main waves already saturate a SIMD's VALU, the prefixes add 1008 packed ops to one of them.)  t0 = tile-relative index of the output's first sample (a multiple of 8); jmax is a multiple
// of 8 (T/2 and D are), so a lane is in or out of a whole 8-tap block.
template <class GeoT>
__device__ __forceinline__ float2 fir_prefix(const float2 *raw, uint32_t t0, uint32_t jmax, const float *h) {
    constexpr uint32_t T = GeoT::T;
    float ar = 0.f, ai = 0.f;
    uint32_t jhi = jmax;                                   // longest prefix in the wave bounds the loop
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)jhi, off); jhi = o > jhi ? o : jhi; }
    float2 xa[8], xb[8]; float ha[8], hb[8];
    auto load = [&](uint32_t jb, float2 *x, float *hh) {   // reads past a lane's own jmax stay inside its T-sample span
        const uint32_t t = t0 + jb;
        const float2 *pp = raw + (t + GeoT::kPad * (t >> GeoT::pshift));
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = pp[i];
        const float4 *hp = reinterpret_cast<const float4 *>(h + jb);
        const float4 h0 = hp[0], h1 = hp[1];
        hh[0] = h0.x; hh[1] = h0.y; hh[2] = h0.z; hh[3] = h0.w; hh[4] = h1.x; hh[5] = h1.y; hh[6] = h1.z; hh[7] = h1.w;
    };
    auto mac = [&](uint32_t jb, const float2 *x, const float *hh) {
        if (jb < jmax) {                                   // whole block in or out (jmax is a multiple of 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) { ar = ar + x[i].x * hh[i]; ai = ai + x[i].y * hh[i]; }
        }
    };
    load(0, xa, ha);
#pragma unroll 1
    for (uint32_t jb = 0; jb < jhi; jb += 16) {            // two 8-tap blocks per trip, the next one always in flight
        if (jb + 8 < T) load(jb + 8, xb, hb);
        mac(jb, xa, ha);
        __builtin_amdgcn_sched_barrier(0);
        if (jb + 16 < T) load(jb + 16, xa, ha);
        mac(jb + 8, xb, hb);
        __builtin_amdgcn_sched_barrier(0);
    }
    return make_float2(ar, ai);
}

// One wave transforms a parked tile (pg windows at fbp, FIR output in rustfft's digit-reversed order) and writes its output:
// LDS operations of one wave execute in order, so the passes are separated by a compiler-level fence only.  twl = the layer
// twiddles (LDS or global), pw0 = the tile's first window.
// freq_levels on transformed windows parked in LDS (one wave): the norms replace the samples in place (every lane reads its share
// first), then one lane per window forms the two sequential half sums (src/fft.rs:95-97).  K: bins per lane the buffer may hold.
template <class GeoT, uint32_t K>
__device__ __forceinline__ void wave_bucket_epilogue_fn(const ChainParams &P, const GeoT &geo, float2 *fbp, uint64_t pw0, uint32_t pg, uint32_t lane) {
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint64_t wrel = pw0 - P.out_window0;
    const uint32_t n_out_s = pg << geo.logW;
    float nm[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) { const uint32_t o = lane + 64 * k; nm[k] = o < n_out_s ? norm_ref(fbp[o]) : 0.f; }
    wsync();
    float *nb = reinterpret_cast<float *>(fbp);
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) { const uint32_t o = lane + 64 * k; if (o < n_out_s) nb[o] = nm[k]; }
    wsync();
    for (uint32_t wl = lane; wl < pg; wl += 64) {                           // (more than 64 windows per buffer: the wave-local kernel at W < 16)
        const float *q = nb + (wl << geo.logW);
        float first = 0.f, second = 0.f;
        for (uint32_t k = 0; k < geo.W / 2; ++k) first = first + q[k];
        for (uint32_t k = geo.W / 2; k < geo.W; ++k) second = second + q[k];
        reinterpret_cast<uint8_t *>(P.out)[wrel + wl] = first < second ? 0 : 1;
    }
}

template <class GeoT, uint32_t KB = 0 /* bucket epilogue: bins per lane the caller's buffer may hold (0: from the geometry) */,
          int PART = 0 /* 0: the whole transform + epilogue; 1: the base butterflies only; 2: the radix-4 layers + epilogue of windows whose base pass is done;
                          3: base butterflies + layers, no epilogue (the caller has its own) */>
__device__ __forceinline__ void wave_fft_epilogue_fn(const ChainParams &P, const GeoT &geo, const float2 *twl, float2 *fbp, uint64_t pw0, uint32_t pg, uint32_t tid) {
    uint32_t lane = tid & 63u;
    asm volatile("" : "+v"(lane));            // opaque: per-lane LDS / output addresses are rebuilt per tile, not hoisted out of the tile loop and spilled
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    const uint32_t base = geo.base_len, log_tpw = geo.logW - geo.log_base;
    const uint32_t n_task = pg << log_tpw;
    if constexpr (PART != 2)
    for (uint32_t t = lane; t < n_task; t += 64) {
        float2 *d = fbp + (size_t)t * base;
        if (base == 16) {
            float2 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = d[i];
            bf16(v, P.tw16_1, P.tw16_2, P.tw16_3, P.root2);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = v[i];
        } else if (base == 8) {
            float2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = d[i];
            bf8(v, P.root2);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = v[i];
        } else if (base == 4) {
            float2 v0 = d[0], v1 = d[1], v2 = d[2], v3 = d[3];
            bf4(v0, v1, v2, v3);
            d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
        } else if (base == 2) {
            float2 v0 = d[0], v1 = d[1];
            bf2(v0, v1);
            d[0] = v0; d[1] = v1;
        }
    }
    if constexpr (PART == 1) return;
    uint32_t cols = base, log_cols = geo.log_base;
    const float2 *tw = twl;
    for (uint32_t layer = 0; layer < geo.layers; ++layer) {
        wsync();
        const uint32_t n_bf = (pg << geo.logW) >> 2;
        for (uint32_t t = lane; t < n_bf; t += 64) {
            const uint32_t chunk = t >> log_cols, i = t & (cols - 1);
            float2 *d = fbp + (size_t)chunk * 4 * cols + i;
            float2 s0 = d[0];
            float2 s1 = cmul(d[cols], tw[3 * i]);
            float2 s2 = cmul(d[2 * cols], tw[3 * i + 1]);
            float2 s3 = cmul(d[3 * cols], tw[3 * i + 2]);
            bf4(s0, s1, s2, s3);
            d[0] = s0; d[cols] = s1; d[2 * cols] = s2; d[3 * cols] = s3;
        }
        tw += 3 * cols;
        cols *= 4;
        log_cols += 2;
    }
    wsync();
    if constexpr (PART == 3) return;
    const uint64_t wrel = pw0 - P.out_window0;
    const uint32_t n_out_s = pg << geo.logW;
    if (P.epi == 2) {
        constexpr uint32_t K = KB ? KB : (GeoT::kFixed ? (GeoT::G_ct * GeoT::W_ct + 63) / 64 : 1);
        wave_bucket_epilogue_fn<GeoT, K>(P, geo, fbp, pw0, pg, lane);
    } else {
        float *outf = reinterpret_cast<float *>(P.out) + (wrel << geo.logW);
        uint8_t *outb = reinterpret_cast<uint8_t *>(P.out) + (wrel << geo.logW);
        for (uint32_t o = lane; o < n_out_s; o += 64) {
            const float2 xv = fbp[o ^ (geo.W >> 1)];
            const float nm = norm_ref(xv);
            if (P.epi == 0) outf[o] = nm;       // (non-temporal stores here measured nothing: 3.343 vs 3.347 ms, round 3)
            else outb[o] = glyph_of<GeoT>(P, nm);
        }
    }
}

// deferred FFT (see k_chain)
template <class GeoT, bool HAS_FIR>
constexpr bool defer_fft_ok(uint32_t nt) {
    if constexpr (!GeoT::kFixed) return false;
    else {
        constexpr uint32_t GW = GeoT::kHalfTile ? GeoT::kHalfOut : GeoT::G * GeoT::W, FL = GeoT::kPackedTile ? GW / 2 : GW;       // lanes the FIR occupies
        return HAS_FIR && (GeoT::kPairFir || GeoT::kPackedTile) && (GeoT::kFlags & kGeoDeferFft) != 0 && GeoT::kBatch == 2 && FL % 64 == 0 && FL + 64 <= nt;
    }
}

// deferred FFT of one long window on four waves (see quad_fft_epilogue)
template <class GeoT, bool HAS_FIR>
constexpr bool quad_fft_ok(uint32_t nt) {
    if constexpr (!GeoT::kFixed) return false;
    else {
        constexpr uint32_t GW = GeoT::kHalfTile ? GeoT::kHalfOut : GeoT::G * GeoT::W, FL = GeoT::kPackedTile ? GW / 2 : GW;
        return defer_fft_ok<GeoT, HAS_FIR>(nt) && GeoT::G == 1 && GeoT::W >= 256 && GeoT::layers >= 1 && GeoT::base_len >= 8 && FL + 4 * 64 <= nt;
    }
}

// fast phase 1 (see k_chain): the tile starts on a row boundary whatever its index and is exactly RCH rows long
template <int FMT, int NT, class GeoT>
constexpr bool fast_p1_ok(int rch, bool whole, bool aligned) {
    if constexpr (!GeoT::kFixed) return false;
    else {
        constexpr uint32_t ROW = NT * FmtTraits<FMT>::SPL;
        constexpr uint32_t tile_raw = GeoT::kHalfTile ? GeoT::kHalfRaw : (GeoT::G - 1) * GeoT::S * GeoT::D + GeoT::W * GeoT::D + GeoT::T;
        constexpr uint32_t step = GeoT::kHalfTile ? GeoT::kHalfOut * GeoT::D : GeoT::S * GeoT::D;      // raw samples between consecutive passes
        // tiles start at (first_window + t G) S D: on a row boundary for every t when G S D is a multiple of the row AND the launch's
        // first window is (the host checks that per launch and sends a misaligned range to the per-sample kernel)
        return whole && aligned && step % ROW == 0 && (GeoT::G * GeoT::S * GeoT::D) % ROW == 0 && (uint32_t)rch == (tile_raw + ROW - 1) / ROW && GeoT::D % FmtTraits<FMT>::SPL == 0 &&
               (GeoT::kFlags & kGeoFastP1);
    }
}

// ---------------------------------------------------------------- the kernel
// RCH / WHOLE: prefetch geometry.
//   WHOLE (rows per tile <= RCH): slot i holds row i of the workgroup's *next* tile; it is refilled
//   right after row i of the current tile has been consumed, so a full tile of loads is in flight
//   during the FIR/FFT phases (prefetch distance = one tile).
//   Chunked (bigger tiles): RCH rows ahead within the tile, the next tile's first chunk at its end.
// LB: waves per SIMD the build is register-budgeted for (__launch_bounds__ 2nd argument).
template <int FMT, int NCO, class GeoT, bool HAS_FIR, int RCH, bool WHOLE, bool ALIGNED, int LB, int NT = kThreads>
__global__ __launch_bounds__(NT, LB) void k_chain(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    constexpr int SPL = FT::SPL;
    constexpr bool HAS_SHIFT = NCO != 0;
    const GeoT geo(P);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *raw = reinterpret_cast<float2 *>(smem);
    // Deferred FFT (FLAGS_ bit 6; packed FIRs that leave at least one wave idle): the FIR keeps G*W/64 (or G*W/128) waves busy and the FFT + epilogue one wave, so
    // the FFT + epilogue of tile i-1 run on an otherwise idle wave WHILE the FIR waves work on tile i (two FFT slots, used
    // alternately).  The FFT is then wave-local — no workgroup barrier between its passes — and a tile costs the workgroup two
    // barriers and phase 1 + FIR of latency instead of four barriers and phase 1 + FIR + FFT + epilogue.
    constexpr bool kDefer = defer_fft_ok<GeoT, HAS_FIR>((uint32_t)NT);
    constexpr bool kQuadFft = quad_fft_ok<GeoT, HAS_FIR>((uint32_t)NT);
    constexpr uint32_t kSlots = GeoT::kBatch;                 // FFT slots in LDS
    constexpr uint32_t kBatch = kDefer ? 1u : GeoT::kBatch;   // tiles per FFT batch
    constexpr uint32_t kLutElems = (FMT == 1 || FMT == 2) ? 256u : 0u;
    float2 *fb0 = raw + geo.lds_raw_elems;                    // kBatch slots of G*W decimated samples (FFT buffers)
    float2 *twl = fb0 + (size_t)kSlots * geo.G * geo.W;       // radix-4 layer twiddles (< W entries), staged once
    float *tapl = reinterpret_cast<float *>(twl + geo.W);     // FIR taps (T floats, padded to a multiple of 4)
    float *lut = tapl + ((GeoT::kBakedTaps && NT == 256) ? 0u : ((geo.T + 3) & ~3u));   // 8-bit unpack table (8-bit formats only); baked taps take no LDS
    // shared-FIR mode (overlapping windows): every decimated output of the tile once + its truncated variant
    float2 *dec = reinterpret_cast<float2 *>(lut + kLutElems);
    float2 *trc = dec + ((geo.G - 1) * geo.S + geo.W);
    // batched FFT: where each parked tile's output goes ({window index relative to out, low / high word; window count})
    uint32_t *bmeta = reinterpret_cast<uint32_t *>(GeoT::kShared ? trc + ((geo.G - 1) * geo.S + geo.W) : dec);
    uint32_t *wq = bmeta + 4 * kSlots;                         // tile queue hand-over: {tile lo, hi} x 2

    const uint32_t tid = threadIdx.x;
    const uint32_t W = geo.W, logW = geo.logW, S = geo.S, D = geo.D, T = geo.T, Dp = geo.Dp;


    // Per-lane NCO constants: 3 doubles per sample slot.  Kernels whose FIR is register-hungry (register-tiled
    // long filters) re-derive them at the top of every tile instead of keeping 8*SPL VGPRs live across the FIR
    // (the table is L2-resident; a tile of such a shape costs tens of thousands of cycles).
    // ... and so do the runtime-geometry kernels (round 4): with the five base-butterfly paths and the generic FIR loops they sit at the
    // 128-VGPR budget of four waves per SIMD, and the lane constants (6 SPL registers) were what spilled to scratch
    constexpr bool kReloadLane = HAS_SHIFT && ((GeoT::kFixed && GeoT::kFirTile > 1) || !GeoT::kFixed);
    constexpr bool kFastP1 = fast_p1_ok<FMT, NT, GeoT>(RCH, WHOLE, ALIGNED);
    constexpr bool kHalf = GeoT::kHalfTile;                 // two passes per window (FixedGeo::kHalfTile); `half` = the pass this iteration runs
    static_assert(!kHalf || (kFastP1 && defer_fft_ok<GeoT, HAS_FIR>((uint32_t)NT)), "half-window tiles: row-aligned phase 1 and the deferred FFT");
    uint32_t half = 0;
    LaneRot lr[SPL];
    auto load_lane_rot = [&](uint32_t first) {
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            uint32_t j = first + u;
            double2 cs = P.jtab[j];
            lr[u].jf = (double)j;
            lr[u].c = cs.x;
            lr[u].s = cs.y;
        }
    };
    if constexpr (HAS_SHIFT && !kReloadLane) load_lane_rot(tid * SPL);
    const uint32_t lane_pad = pad_index(geo, tid * SPL);      // LDS element of this lane's first sample in a row
    // Stage the layer twiddles in LDS: phases 2-4 then issue no vector-memory loads, so nothing in
    // them has to wait behind the next tile's prefetch (vmcnt retires in order).
    {
        const uint32_t n_tw = W - geo.base_len;               // 3*(base + 4*base + ...) = W - base
        for (uint32_t i = tid; i < n_tw; i += NT) twl[i] = P.tw[i];
        if (tid == 0) bmeta[0] = 0;                           // arrival counter of the four-wave deferred FFT (see quad_fft_epilogue)
        // ... and the taps: the FIR loop then contains LDS reads only, so its waits are counted
        // lgkmcnt(N) instead of a full drain per batch (scalar loads share that counter and return
        // out of order, which forces lgkmcnt(0)).
        if constexpr (!(GeoT::kBakedTaps && NT == 256)) { for (uint32_t i = tid; i < T; i += NT) tapl[i] = P.taps[i]; }
    }
    __syncthreads();

    const uint64_t n_tiles = (P.n_windows + geo.G - 1) / geo.G;

    // Register prefetch pipeline.  Slots whose row does not exist in the tile re-load its last
    // row (an L2 hit) so that every load stays unconditional.
    // Tile walk.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2.
    // Neighbouring tiles share their edge row (a tile is not a whole number of rows) and the FIR halo, so each
    // XCD walks one contiguous eighth of the tile range: its workgroups sit on neighbouring tiles at the same
    // time and the shared rows are L2 hits instead of a second HBM fetch.
    uint64_t walk_base = 0, walk_local = blockIdx.x, walk_step = gridDim.x, walk_limit = n_tiles;
    const bool xcd_walk = (gridDim.x & 7u) == 0 && n_tiles >= 8;
    const uint64_t n8 = (n_tiles + 7) / 8;
    auto xcd_limit = [&](uint32_t x) -> uint64_t { const uint64_t b = (uint64_t)x * n8; return b >= n_tiles ? 0 : (n_tiles - b < n8 ? n_tiles - b : n8); };
    if (xcd_walk) {
        walk_base = (uint64_t)(blockIdx.x & 7u) * n8;
        walk_local = blockIdx.x >> 3;
        walk_step = gridDim.x >> 3;
        walk_limit = xcd_limit(blockIdx.x & 7u);
    }
    auto walk_tile = [&](uint64_t local) -> uint64_t { return local < walk_limit ? walk_base + local : n_tiles; };   // n_tiles = none
    // Dynamic tile queue (P.work): instead of a fixed stride, a workgroup CLAIMS its tiles — one atomic add on the counter of
    // its XCD group's eighth of the tile range, issued a tile ahead so the reply is never waited for, and on the other
    // groups' counters once its own eighth is exhausted.  Workgroups of one XCD still sit on neighbouring tiles (shared halo
    // rows are L2 hits), but none idles while another still has a backlog: with the static walk the fastest workgroups
    // finish at 80 % of the kernel's span (per-CU and per-XCD speed differs by a few %) and ~10 % of the chip-time is lost.
    const bool dyn = P.work != nullptr && xcd_walk;
    const uint32_t my_x = blockIdx.x & 7u;
    auto claim_resolve = [&](unsigned long long got) -> uint64_t {        // one lane; `got` = reply of the add on the own counter
        // The group index is made opaque here: left visible, hipcc unrolls the help-the-others loop, hoists its seven limits,
        // bases and counter addresses out of the TILE loop and spills them (56 of the kernel's SGPR spills, each read back with
        // a v_readlane on the FIR wave between its last tap and the barrier the whole workgroup waits at).
        uint32_t mx = my_x;
        asm volatile("" : "+s"(mx));
        // every workgroup walks its first TWO tiles statically (no atomic round trip and no hand-over barriers in front of the
        // first loads); the counters hand out what lies behind those two rounds of each group
        const uint64_t dyn_base = 2 * walk_step;
        if (got + dyn_base < xcd_limit(mx)) return (uint64_t)mx * n8 + dyn_base + got;
#pragma unroll 1
        for (uint32_t k = 1; k < 8; ++k) {                                // own eighth exhausted: help the others
            const uint32_t xx = (mx + k) & 7u;
            const uint64_t lim = xcd_limit(xx);
            if (lim <= dyn_base) continue;
            const unsigned long long i = atomicAdd(&P.work[16 * xx], 1ull);
            if (i + dyn_base < lim) return (uint64_t)xx * n8 + dyn_base + i;
        }
        return n_tiles;
    };
    uint64_t tile, tile_n;                                                // this tile and the next one (wave-uniform)
    tile = walk_tile(walk_local);                                         // static for the first two rounds, dynamic or static after
    tile_n = walk_tile(walk_local + walk_step);
    TileGeo tg = tile_geo<FMT, NT>(P, geo, tile, n_tiles);
    Vec pf[RCH];
    if (tg.valid) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) pf[i] = fetch_row<FMT, NT, ALIGNED, (GeoT::kFlags & kGeoNtLoads) != 0>(P, tg, (uint32_t)i < tg.n_rows ? i : tg.n_rows - 1, tid);
    }

#ifdef QD_WGTIME       // diagnostic build: when each workgroup started / finished (100 MHz realtime counter) and on which XCD
    unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    double rt_touch = 0.0;   // keeps the row-table L2 prefetch loads alive (see prefetch_rowtab)
    // deferred FFT: the previous tile, parked in FFT slot dpar ^ 1
    uint32_t dpar = 0, dprev_gcnt = 0;
    uint64_t dprev_w0 = 0;
    bool dprev_valid = false;
    // One wave transforms the parked tile and writes its output: LDS operations of one wave execute in order, so the passes
    // are separated by a compiler-level fence only.
    auto wave_fft_epilogue = [&](float2 *fbp, uint64_t pw0, uint32_t pg) { wave_fft_epilogue_fn<GeoT>(P, geo, twl, fbp, pw0, pg, tid); };
    // The same on FOUR waves, for one long window (W >= 256): Radix4's last layer combines four contiguous sub-transforms of W/4
    // points, and everything below it stays inside one sub-transform — so wave v transforms quarter v on its own (no workgroup
    // barrier, the FIR waves run on), the four meet ONCE at an arrival counter in LDS, and each then takes a quarter of the last
    // layer's butterflies: butterfly i yields bins i, i + W/4, i + W/2, i + 3W/4, which after the fftshift are the outputs at the
    // same four places in rotated order, so norms / glyph codes go straight from registers to HBM.  A 1024-point window costs
    // ~1/4 of the single-wave latency — short enough to hide under the FIR (cfg4), where the single wave was the long pole.
    uint32_t quad_gen = 0;                                                // syncs passed so far (uniform over the four waves)
    auto quad_fft_epilogue = [&](float2 *fbp, uint64_t pw0, uint32_t v) {
        uint32_t lane = tid & 63u;
        asm volatile("" : "+v"(lane));
        auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
        const uint32_t Wq = geo.W >> 2, base = geo.base_len;
        float2 *sub = fbp + (size_t)v * Wq;
        for (uint32_t t = lane; t < Wq / base; t += 64) {
            float2 *d = sub + (size_t)t * base;
            if (base == 16) {
                float2 x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = d[i];
                bf16(x, P.tw16_1, P.tw16_2, P.tw16_3, P.root2);
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = x[i];
            } else {
                float2 x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = d[i];
                bf8(x, P.root2);
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = x[i];
            }
        }
        uint32_t cols = base, log_cols = geo.log_base;
        const float2 *tw = twl;
        for (uint32_t layer = 0; layer + 1 < geo.layers; ++layer) {
            wsync();
            for (uint32_t t = lane; t < Wq / 4; t += 64) {
                const uint32_t chunk = t >> log_cols, i = t & (cols - 1);
                float2 *d = sub + (size_t)chunk * 4 * cols + i;
                float2 s0 = d[0];
                float2 s1 = cmul(d[cols], tw[3 * i]);
                float2 s2 = cmul(d[2 * cols], tw[3 * i + 1]);
                float2 s3 = cmul(d[3 * cols], tw[3 * i + 2]);
                bf4(s0, s1, s2, s3);
                d[0] = s0; d[cols] = s1; d[2 * cols] = s2; d[3 * cols] = s3;
            }
            tw += 3 * cols;
            cols *= 4;
            log_cols += 2;
        }
        // the four waves meet: this wave's stores are issued before its arrival (LDS executes one wave's operations in order), and
        // nothing after the wait is read early
        ++quad_gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&bmeta[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&bmeta[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * quad_gen) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // last layer (cols == W/4) + epilogue, butterflies [v W/16, (v+1) W/16)
        const uint64_t wrel = pw0 - P.out_window0;
        float *outf = reinterpret_cast<float *>(P.out) + (wrel << geo.logW);
        uint8_t *outb = reinterpret_cast<uint8_t *>(P.out) + (wrel << geo.logW);
        for (uint32_t t = lane; t < Wq / 4; t += 64) {
            const uint32_t i = v * (Wq / 4) + t;
            const float2 *d = fbp + i;
            float2 s0 = d[0];
            float2 s1 = cmul(d[Wq], tw[3 * i]);
            float2 s2 = cmul(d[2 * Wq], tw[3 * i + 1]);
            float2 s3 = cmul(d[3 * Wq], tw[3 * i + 2]);
            bf4(s0, s1, s2, s3);
            // output o = i + k W/4 is bin o ^ (W/2) = i + ((k + 2) & 3) W/4
            const float n0 = norm_ref(s2), n1 = norm_ref(s3), n2 = norm_ref(s0), n3 = norm_ref(s1);
            if (P.epi == 0) { outf[i] = n0; outf[i + Wq] = n1; outf[i + 2 * Wq] = n2; outf[i + 3 * Wq] = n3; }
            else {
                outb[i] = glyph_of<GeoT>(P, n0); outb[i + Wq] = glyph_of<GeoT>(P, n1);
                outb[i + 2 * Wq] = glyph_of<GeoT>(P, n2); outb[i + 3 * Wq] = glyph_of<GeoT>(P, n3);
            }
        }
    };
    uint32_t bslot = 0;      // parked tiles of the current FFT batch (wave-uniform)
    QD_STAMP_DECL
    QD_STAMP_START();
    while (tg.valid) {
        const uint64_t w0 = tg.w0;
        const uint32_t g_cnt = tg.g_cnt;

        // ---------------- phase 1: HBM -> unpack -> NCO -> LDS
        double rt_pf = 0.0;
        // Wave priority: the arithmetic phases sit on the workgroup's barrier chain, phase 1 mostly waits for HBM,
        // so a wave in FIR (1) / FFT + epilogue (3) is issued ahead of the other workgroups' phase-1 waves (0): the
        // closer a workgroup is to finishing its tile, the sooner it gets issue slots (cfg3' +7 %, cfg2 +2-4 %;
        // scripts/prio_probe.py).
        __builtin_amdgcn_s_setprio(0);
        if constexpr (kReloadLane) {
            uint32_t first = tid * SPL;
            asm volatile("" : "+v"(first));          // opaque per tile: the loads are not hoisted out of the tile loop
            load_lane_rot(first);
        }
        unsigned long long claim = 0;
        if (dyn && tid == 0 && (!kHalf || half == 1)) claim = atomicAdd(&P.work[16 * my_x], 1ull);     // the tile after next; the reply is read after the FIR
        if constexpr (kFastP1) {
            // Row-aligned tiles (the tile stride G*S*D a multiple of the row length, the tile a compile-time number of rows): every row
            // offset is an immediate, the loads go through a per-tile buffer descriptor whose hardware range check replaces the clamp
            // arithmetic (a vector past the slab's end reads as zero and is never used), interior rows carry no bounds logic and only
            // the last row is predicated, with a compile-time extent.  A SHORT last tile (n_windows not a multiple of G) runs in the
            // same launch: it loads and parks the full row count (the descriptor covers the slab end, the host extends the row table
            // by one tile) and the FIR / epilogue are bounded by g_cnt.  ~250 fewer scalar / vector instructions per tile and wave
            // than the general path below.
            constexpr uint32_t ROWB = NT * SPL * FT::BPS, VECB = SPL * FT::BPS;
            constexpr uint32_t kTileRaw = kHalf ? GeoT::kHalfRaw : (GeoT::G - 1) * GeoT::S * GeoT::D + GeoT::W * GeoT::D + GeoT::T;
            constexpr uint32_t kRem = kTileRaw - (RCH - 1) * (NT * SPL);          // samples in the last row
            constexpr uint32_t kHalfStep = GeoT::kHalfOut * GeoT::D;               // raw samples from a window's first pass to its second
            uint64_t tile_pf = tile_n < n_tiles ? tile_n : tile;                   // last tile of this workgroup: harmless re-loads
            if (kHalf && half == 0) tile_pf = tile;                                // the next pass is this window's second half
            const uint64_t ns_n = (P.first_window + tile_pf * GeoT::G) * ((uint64_t)GeoT::S * GeoT::D) + ((kHalf && half == 0) ? kHalfStep : 0u);
            const uint64_t left = (P.src_first + P.src_count - ns_n) * FT::BPS;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns_n - P.src_first) * FT::BPS, 0,
                                                                left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
            if constexpr (HAS_SHIFT) {
                uint32_t r = tid < (uint32_t)RCH ? tid : (uint32_t)RCH - 1;
                asm volatile("" : "+v"(r));
                rt_pf = P.rowtab[ns_n / (NT * SPL) - P.rowtab_row0 + r].c;          // L2 touch of the next tile's row bases (see prefetch_rowtab)
            }
            const_f64_p rows = (const_f64_p)(uintptr_t)(P.rowtab + (tg.r0 + (kHalf ? half * (kHalfStep / (NT * SPL)) : 0u) - P.rowtab_row0));
            RowBase rb_next{};
            if constexpr (HAS_SHIFT) rb_next = load_rowbase_at(rows, 0);
            TileGeo gl = tg;
            gl.tile_raw = kTileRaw;
#pragma unroll
            for (int i = 0; i < RCH; ++i) {
                const Vec v = pf[i];
                const RowBase rb = rb_next;
                if constexpr (HAS_SHIFT) { if (i + 1 < RCH) rb_next = load_rowbase_at(rows, i + 1); }
                if (i + 1 < RCH || kRem == NT * SPL) {
                    process_row<FMT, NT, NCO, true>(P, geo, gl, i * (NT * SPL), tid, v, rb, lr, lane_pad, lut, raw);
                } else {
                    if (tid * SPL < kRem) process_row<FMT, NT, NCO, (kRem % SPL) == 0>(P, geo, gl, i * (NT * SPL), tid, v, rb, lr, lane_pad, lut, raw);
                }
                __builtin_amdgcn_sched_barrier(0);       // refill slot i only after row i is consumed (see below)
                typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                // lanes past the tile's edge in the last row collapse onto its last needed vector (one cache line instead of a
                // row of bytes this tile never uses)
                constexpr uint32_t kLastVec = ((kRem * FT::BPS - 1) / VECB) * VECB;
                uint32_t voff = tid * VECB;
                if (i + 1 == RCH && kRem != NT * SPL) voff = voff < kLastVec ? voff : kLastVec;
                // the tile's first and last row are read by the neighbouring tile as well (its halo): with bit 16 they keep the default
                // cache policy and stay L2 hits for the second reader; the rows in between are read once and go non-temporal
                const bool shared_row = (GeoT::kFlags & kGeoNtInner) && (i == 0 || i + 1 == RCH);
                if constexpr (sizeof(Vec) == 16) {
                    const v4u_t w = shared_row ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(i * ROWB), 0)
                                               : __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(i * ROWB), ct_load_aux(GeoT::kFlags));
                    pf[i].x = w.x; pf[i].y = w.y; pf[i].z = w.z; pf[i].w = w.w;
                } else {
                    const v2u_t w = shared_row ? __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(i * ROWB), 0)
                                               : __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(i * ROWB), ct_load_aux(GeoT::kFlags));
                    pf[i].x = w.x; pf[i].y = w.y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        if constexpr (WHOLE) {
            TileGeo ng = tile_geo<FMT, NT>(P, geo, tile_n, n_tiles);
            if (!ng.valid) ng = tg;                  // last tile of this workgroup: harmless re-loads
            rt_pf = prefetch_rowtab<HAS_SHIFT>(P, ng, tid);
            RowBase rb_next{};
            if constexpr (HAS_SHIFT) rb_next = load_rowbase(P, tg.r0);
#pragma unroll
            for (int i = 0; i < RCH; ++i) {
                const Vec v = pf[i];
#ifdef QD_STAMP
                asm volatile("" :: "v"(v.x), "v"(v.y));      // force the wait for this row's data here
                QD_STAMP_ROW(0);
#endif
                if ((uint32_t)i < tg.n_rows) {
                    const RowBase rb = rb_next;
#ifdef QD_STAMP
                    asm volatile("" :: "s"(rb.c), "s"(rb.nf));
                    QD_STAMP_ROW(2);
#endif
                    if constexpr (HAS_SHIFT) {           // scalar load for the next row while this one computes
                        if ((uint32_t)i + 1 < tg.n_rows) rb_next = load_rowbase(P, tg.r0 + i + 1);
                    }
                    process_row_any<FMT, NT, NCO>(P, geo, tg, i, tid, v, rb, lr, lane_pad, lut, raw);
                    QD_STAMP_ROW(3);
                }
                // Refill slot i only now, into the registers row i just vacated.  Issued before the row is consumed
                // the new load needs different registers, the slots rotate by one per tile, and hipcc squares that at the
                // loop's back edge with register copies behind an s_waitcnt vmcnt(0) — a full drain of the next tile's
                // prefetch at the end of every tile.
                __builtin_amdgcn_sched_barrier(0);
                pf[i] = fetch_row<FMT, NT, ALIGNED, (GeoT::kFlags & kGeoNtLoads) != 0>(P, ng, (uint32_t)i < ng.n_rows ? i : ng.n_rows - 1, tid);
                QD_STAMP_ROW(1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (uint32_t r = 0; r < tg.n_rows; r += RCH) {
                Vec cur[RCH];
#pragma unroll
                for (int i = 0; i < RCH; ++i) cur[i] = pf[i];
                TileGeo ng = tg;
                uint32_t rbase = r + RCH;
                if (rbase >= tg.n_rows) {
                    const TileGeo t2 = tile_geo<FMT, NT>(P, geo, tile_n, n_tiles);
                    if (t2.valid) { ng = t2; rbase = 0; rt_pf = prefetch_rowtab<HAS_SHIFT>(P, t2, tid); } else { rbase = tg.n_rows - 1; }
                }
#pragma unroll
                for (int i = 0; i < RCH; ++i)
                    pf[i] = fetch_row<FMT, NT, ALIGNED, (GeoT::kFlags & kGeoNtLoads) != 0>(P, ng, rbase + i < ng.n_rows ? rbase + i : ng.n_rows - 1, tid);
                RowBase rb[RCH];
#pragma unroll
                for (int i = 0; i < RCH; ++i) {
                    rb[i] = RowBase{};
                    if constexpr (HAS_SHIFT) { if (r + i < tg.n_rows) rb[i] = load_rowbase(P, tg.r0 + r + i); }
                }
#pragma unroll
                for (int i = 0; i < RCH; ++i)
                    if (r + i < tg.n_rows) process_row_any<FMT, NT, NCO>(P, geo, tg, r + i, tid, cur[i], rb[i], lr, lane_pad, lut, raw);
            }
        }
        QD_STAMP_AT(0);
        __syncthreads();
        QD_STAMP_AT(1);

        __builtin_amdgcn_s_setprio(1);
        if constexpr (kDefer) {
            constexpr uint32_t GW = GeoT::G_ct * GeoT::W_ct, GWP = kHalf ? GeoT::kHalfOut : GW, FL = GeoT::kPackedTile ? GWP / 2 : GWP;      // GWP: outputs of one pass
            float2 *fb_cur = fb0 + (size_t)dpar * GW, *fb_prev = fb0 + (size_t)(dpar ^ 1u) * GW;
            const uint32_t n_out_d = kHalf ? GWP : (g_cnt << logW);
            const uint32_t log_width_d = 2 * geo.layers;
            const uint32_t k_first = kHalf ? half * GWP : 0u;  // first output of this pass; the raw buffer starts at ITS first sample
            if (tid < FL) {                                    // the FIR waves
                if constexpr (GeoT::kPackedTile) {             // two outputs per lane, truncated ones as in-chain snapshots
                    const uint32_t o0 = tid * 2;
                    if (o0 < n_out_d) {
                        const uint32_t g = kHalf ? 0u : (o0 >> logW), k0 = kHalf ? k_first + o0 : (o0 & (W - 1));
                        uint32_t jm[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) { const uint32_t j2 = (W - (k0 + r)) * D + T / 2; jm[r] = j2 < T ? j2 : T; }
                        float2 full[2], snp[2];
                        const uint32_t q0 = kHalf ? o0 : g * S + k0;
                        if ((tid >> 8) & 1u) fir_tiled2_pk<GeoT, 1>(raw + (q0 * D + GeoT::kPad * (q0 / 2)), tapl, full, jm, snp); else fir_tiled2_pk<GeoT, 0>(raw + (q0 * D + GeoT::kPad * (q0 / 2)), tapl, full, jm, snp);
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const uint32_t k = k0 + r;
                            const uint32_t xx = k & ((1u << log_width_d) - 1), yy = k >> log_width_d;
                            fb_cur[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = jm[r] < T ? snp[r] : full[r];
                        }
                    }
                } else if (tid < n_out_d) {                    // one lane per complex output
                    const uint32_t g = kHalf ? 0u : (tid >> logW), k = kHalf ? k_first + tid : (tid & (W - 1));
                    uint32_t jmax = (W - k) * D + T / 2;
                    if (jmax > T) jmax = T;
                    const float2 *rp = raw + (size_t)((kHalf ? tid : g * S + k) + geo.a0) * Dp;
                    const float2 v = QD_DBG(P, 2) ? rp[0] : fir_pair<GeoT>(rp, jmax, tapl);      // dbg: timing-only ablation
                    const uint32_t xx = k & ((1u << log_width_d) - 1), yy = k >> log_width_d;
                    fb_cur[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = v;
                }
            } else if (kHalf && half != 0) {                   // the previous window was transformed beside this window's first pass
            } else if (kQuadFft && P.epi != 2) {               // one long window: four spare waves share its FFT + epilogue
                if ((tid >> 6) < FL / 64 + 4 && dprev_valid) {
                    __builtin_amdgcn_s_setprio(3);
                    quad_fft_epilogue(fb_prev, dprev_w0, (tid >> 6) - FL / 64);
                }
            } else if ((tid >> 6) == FL / 64 && dprev_valid) { // the first spare wave: previous tile's FFT + epilogue
                __builtin_amdgcn_s_setprio(3);                 // one wave's serial work next to FIR waves on its SIMD: it goes first
                wave_fft_epilogue(fb_prev, dprev_w0, dprev_gcnt);
            }
            QD_STAMP_AT(2);
            if constexpr (kHalf) {
                if (half == 0) {                               // second pass of the same window next: no tile advance
                    __syncthreads();                           // the raw buffer is free, fb_prev consumed
                    QD_STAMP_AT(3);
                    half = 1;
                    rt_touch += rt_pf;
                    continue;
                }
                half = 0;
            }
            if (dyn && tid == 0) {
                const uint64_t t2 = claim_resolve(claim);
                wq[0] = (uint32_t)t2; wq[1] = (uint32_t)(t2 >> 32);
            }
            const TileGeo tg_nx = tile_geo<FMT, NT>(P, geo, tile_n, n_tiles);
            __syncthreads();                                   // fb_cur complete, fb_prev consumed, the raw tile free
            QD_STAMP_AT(3);
            dprev_w0 = w0; dprev_gcnt = g_cnt; dprev_valid = true; dpar ^= 1u;
            tile = tile_n;
            if (dyn) tile_n = ((uint64_t)__builtin_amdgcn_readfirstlane(wq[1]) << 32) | __builtin_amdgcn_readfirstlane(wq[0]);
            else { walk_local += walk_step; tile_n = walk_tile(walk_local + walk_step); }
            rt_touch += rt_pf;
            tg = tg_nx;
            QD_STAMP_TILE();
            continue;
        }
        // ---------------- phase 2: FIR + decimate (or plain window gather), scatter for the FFT
        float2 *fb = fb0 + (size_t)bslot * geo.G * geo.W;     // this tile's slot of the FFT batch
        const uint32_t n_out = g_cnt << logW;
        const uint32_t log_width = 2 * geo.layers;   // width = W / base_len = 4^layers
        const bool cf32_out = !GeoT::kFixed && P.epi == 3;   // QD_EPI_CF32_BLOCKS (generic kernels only)
        // Overlapping windows (S < W): the same decimated sample q sits in W/S windows.  Its value is
        // the same in all of them except the one where it falls in the truncated tail, and
        // that truncated value is a prefix of the same chain.  So compute each q ONCE (full chain +
        // snapshot) and let the windows gather — W/S times less FIR work than per-window evaluation.
        // Valid when a sample is truncated in at most one window: ntrunc <= S.
        const uint32_t c_half = T - T / 2;
        const uint32_t ntrunc = c_half ? (c_half + D - 1) / D - 1 : 0;
        // compile-time for shape-specialised kernels (so only ONE FIR variant is instantiated and the long
        // unrolled chain stays in registers), a wave-uniform runtime flag for the generic ones
        bool shared;
        if constexpr (GeoT::kFixed) shared = GeoT::kShared; else shared = HAS_FIR && S < W && ntrunc <= S && !QD_DBG(P, 2) && P.epi != 3;
        if constexpr (!GeoT::kFixed || GeoT::kShared) if (shared) {
            const uint32_t Q = (g_cnt - 1) * S + W;
            if constexpr (GeoT::kFixed && GeoT::kFirTile > 1) {
                constexpr int R = (int)GeoT::kFirTile;
                for (uint32_t q0 = tid * R; q0 < Q; q0 += NT * R) {
                    uint32_t jm[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t qi = q0 + r;
                        jm[r] = T;
                        if (qi + ntrunc >= W) {
                            const uint32_t g = (qi - (W - ntrunc)) / S, k = qi - g * S;
                            if (k < W && g < g_cnt) { const uint32_t j2 = (W - k) * D + T / 2; if (j2 < T) jm[r] = j2; }
                        }
                    }
                    float2 full[R], snp[R];
                    fir_tiled<R, GeoT>(raw + (q0 * D + GeoT::kPad * (q0 / R)), jm, tapl, full, snp);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (q0 + r < Q) { dec[q0 + r] = full[r]; if (jm[r] < T) trc[q0 + r] = snp[r]; }
                    }
                }
            } else if constexpr (GeoT::split_ok_shared((uint32_t)NT)) {
                for (uint32_t t = tid; t < 2u * Q; t += NT) {
                    const uint32_t qi = t >> 1, part = t & 1u;
                    uint32_t jmax = T;
                    if (qi + ntrunc >= W) {
                        const uint32_t g = (qi - (W - ntrunc)) / S, k = qi - g * S;
                        if (k < W && g < g_cnt) { const uint32_t jm = (W - k) * D + T / 2; if (jm < T) jmax = jm; }
                    }
                    float snapv = 0.f;
                    const float full = fir_comp<GeoT>(reinterpret_cast<const float *>(raw + (size_t)(qi + geo.a0) * Dp) + part, jmax, tapl, &snapv);
                    reinterpret_cast<float *>(dec + qi)[part] = full;
                    if (jmax < T) reinterpret_cast<float *>(trc + qi)[part] = snapv;
                }
            } else
            for (uint32_t qi = opaque_u32(tid); qi < Q; qi += NT) {      // opaque start: per-lane LDS addresses are recomputed per tile, not hoisted and spilled
                uint32_t jmax = T;
                if (qi + ntrunc >= W) {                      // may be in the truncated tail of window g
                    const uint32_t g = (qi - (W - ntrunc)) / S, k = qi - g * S;
                    if (k < W && g < g_cnt) { const uint32_t jm = (W - k) * D + T / 2; if (jm < T) jmax = jm; }
                }
                const float2 *rowp = raw + (size_t)(qi + geo.a0) * Dp;
                float accr = 0.f, acci = 0.f;
                float2 snap = make_float2(0.f, 0.f);
                if constexpr (GeoT::kUnrolledShared) {
                    // straight-line, pinned tap loop (taps may be immediates: FLAGS_ bit 1), two scalar chains per lane
                    const float2 full = fir_pair<GeoT, !GeoT::baked_request>(rowp, jmax, tapl, &snap);
                    accr = full.x; acci = full.y;
                } else if constexpr (GeoT::kFixed) {
                    fir_span<false, GeoT, true>(geo, rowp, geo.b0, 0, T, jmax, tapl, accr, acci, &snap);
                } else {
                    fir_span<false>(geo, rowp, geo.b0, 0, T, T, tapl, accr, acci);
                    if (jmax < T) {                          // generic kernels: second, predicated pass for the prefix
                        float sr = 0.f, si = 0.f;
                        fir_span<true>(geo, rowp, geo.b0, 0, T, jmax, tapl, sr, si);
                        snap = make_float2(sr, si);
                    }
                }
                dec[qi] = make_float2(accr, acci);
                if (jmax < T) trc[qi] = snap;
            }
            __syncthreads();
            uint32_t o_first = tid;
            asm volatile("" : "+v"(o_first));
            for (uint32_t o = o_first; o < n_out; o += NT) {
                const uint32_t g = o >> logW, k = o & (W - 1);
                const uint32_t qi = g * S + k;
                const bool tr = (W - k) * D + T / 2 < T;
                const float2 v = tr ? trc[qi] : dec[qi];
                const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                fb[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = v;
            }
        }
        constexpr bool kSplit = HAS_FIR && GeoT::split_ok((uint32_t)NT);
        if constexpr (kSplit) {
            constexpr bool kPlanarK = GeoT::kPlanar && NT == 256;
            const uint32_t t_end = kPlanarK ? 2u * geo.G * geo.W : 2u * n_out;
            for (uint32_t t = tid; t < t_end; t += NT) {
                // interleaved tile: lanes alternate re / im of one output; planar tile: the first G*W lanes take the re chains,
                // the next G*W the im chains, so a wave reads ONE plane at a lane stride of DpP floats (conflict-free b128)
                uint32_t o = t >> 1, part = t & 1u;
                if constexpr (GeoT::kPlanar && NT == 256) { o = t & (GeoT::G * GeoT::W - 1); part = t / (GeoT::G * GeoT::W); if (o >= n_out) continue; }
                const uint32_t g = o >> logW, k = o & (W - 1);
                uint32_t jmax = (W - k) * D + T / 2;
                if (jmax > T) jmax = T;
                float v;
                if constexpr (GeoT::kPlanar && NT == 256) {
                    const float *xp = reinterpret_cast<const float *>(raw) + part * GeoT::plane_floats + (size_t)(g * S + k + geo.a0) * GeoT::DpP;
                    v = QD_DBG(P, 2) ? xp[0] : fir_comp_planar<GeoT>(xp, jmax, tapl);
                } else {
                    const float *xp = reinterpret_cast<const float *>(raw + (size_t)(g * S + k + geo.a0) * Dp) + part;
                    v = QD_DBG(P, 2) ? xp[0] : fir_comp<GeoT>(xp, jmax, tapl);     // dbg: timing-only ablation
                }
                const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                reinterpret_cast<float *>(fb + (g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base))[part] = v;
            }
        } else
        if constexpr (GeoT::kFixed && !GeoT::kShared && GeoT::kFirTile > 1 && HAS_FIR && GeoT::helper_ok((uint32_t)NT) && !GeoT::kPackedTile) {
            // main lanes: R outputs each, no truncation logic; spare waves: the truncated tails as direct prefixes
            constexpr int R = (int)GeoT::kFirTile;
            constexpr uint32_t NTR = GeoT::kNtrunc, n_main_max = GeoT::G * GeoT::W / R;
            if (tid < n_main_max) {
                const uint32_t o0 = tid * R;
                if (o0 < n_out) {
                    const uint32_t g = o0 >> logW, k0 = o0 & (W - 1);
                    uint32_t jm[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) jm[r] = T;
                    float2 full[R], snp[R];
                    { const uint32_t q0 = g * S + k0; fir_tiled<R, GeoT>(raw + (q0 * D + GeoT::kPad * (q0 / R)), jm, tapl, full, snp); }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t k = k0 + r;
                        if (k + NTR < W) {                               // the truncated ones belong to the helper lanes
                            const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                            fb[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = full[r];
                        }
                    }
                }
            } else {
                const uint32_t hidx = tid - n_main_max;
                if (NTR > 0 && hidx < g_cnt * NTR) {
                    // The helper wave's serial prefixes are as long as a main wave's whole share and it sits on a SIMD with two
                    // main waves: at equal priority it gets a third of the issue slots and the whole workgroup waits for it.
                    __builtin_amdgcn_s_setprio(3);
                    const uint32_t g = hidx / NTR, k = W - NTR + hidx % NTR;
                    const uint32_t jmax = (W - k) * D + T / 2;           // < T for these k
                    const float2 v = fir_prefix<GeoT>(raw, (g * S + k) * D + GeoT::c, jmax, tapl);
                    const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                    fb[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = v;
                }
            }
        } else
        if constexpr (GeoT::kFixed && !GeoT::kShared && GeoT::kFirTile > 1 && HAS_FIR) {
            constexpr int R = (int)GeoT::kFirTile;
            for (uint32_t o0 = tid * R; o0 < n_out; o0 += NT * R) {
                const uint32_t g = o0 >> logW, k0 = o0 & (W - 1);          // W % R == 0: the R outputs share a window
                uint32_t jm[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { const uint32_t j2 = (W - (k0 + r)) * D + T / 2; jm[r] = j2 < T ? j2 : T; }
                float2 full[R], snp[R];
                {
                    const uint32_t q0 = g * S + k0;
                    if constexpr (GeoT::kPackedTile) { if ((tid >> 8) & 1u) fir_tiled2_pk<GeoT, 1>(raw + (q0 * D + GeoT::kPad * (q0 / R)), tapl, full, jm, snp); else fir_tiled2_pk<GeoT, 0>(raw + (q0 * D + GeoT::kPad * (q0 / R)), tapl, full, jm, snp); }
                    else fir_tiled<R, GeoT>(raw + (q0 * D + GeoT::kPad * (q0 / R)), jm, tapl, full, snp);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t k = k0 + r;
                    const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                    fb[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = jm[r] < T ? snp[r] : full[r];
                }
            }
        } else
        if constexpr (HAS_FIR && GeoT::kPairFir) {
            for (uint32_t o = tid; o < n_out; o += NT) {
                const uint32_t g = o >> logW, k = o & (W - 1);
                uint32_t jmax = (W - k) * D + T / 2;
                if (jmax > T) jmax = T;
                const float2 *rp = raw + (size_t)(g * S + k + geo.a0) * Dp;
                const float2 v = QD_DBG(P, 2) ? rp[0] : fir_pair<GeoT>(rp, jmax, tapl);      // dbg: timing-only ablation
                const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                fb[(g << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = v;
            }
        } else
        if constexpr (!GeoT::kFixed || !GeoT::kShared) if (!shared)
        for (uint32_t o = tid; o < n_out; o += NT) {
            const uint32_t g = o >> logW, k = o & (W - 1);
            const uint32_t q = g * S + k;
            float accr = 0.f, acci = 0.f;
            if (HAS_FIR && !QD_DBG(P, 2)) {
                // jmax(k) = min(T, valid - (k*D + c)) with valid = B*D + T (full read of a block of B outputs;
                // B == W for the FFT sinks, B = 0x1000 > W for the write sink whose tiles are sub-blocks)
                const uint32_t kb = cf32_out ? (((uint32_t)(w0 + g)) & P.blk_sub_mask) * W + k : k;
                uint32_t jmax = ((cf32_out ? P.blk_len : W) - kb) * D + T / 2;
                if (jmax > T) jmax = T;
                const float2 *rowp = raw + (size_t)(q + geo.a0) * Dp;
                if (geo.T_fast == T || __all(jmax == T)) {
                    fir_span<false>(geo, rowp, geo.b0, 0, T, T, tapl, accr, acci);
                } else if constexpr (GeoT::kFixed) {
                    fir_span<false, GeoT, true>(geo, rowp, geo.b0, 0, T, jmax, tapl, accr, acci);
                } else {
                    fir_span<false>(geo, rowp, geo.b0, 0, geo.T_fast, T, tapl, accr, acci);
                    const float2 *rowp1 = raw + (size_t)(q + geo.a1) * Dp;
                    fir_span<true>(geo, rowp1, geo.b1, geo.T_fast, T, jmax, tapl, accr, acci);
                }
            } else {
                float2 v = raw[HAS_FIR ? q * Dp : q];
                accr = v.x; acci = v.y;
                if constexpr (!GeoT::kFixed && !HAS_FIR) {
                    if (P.window) { const float wv = P.window[k]; accr = accr * wv; acci = acci * wv; }   // Complex<f32> * f32
                }
            }
            // bitreversed_transpose::<4>(base_len, ..): out[y + rev(x)*base] = in[x + y*width]
            const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
            const uint32_t pos = cf32_out ? k : yy + (rev4(xx, geo.layers) << geo.log_base);   // write sink: natural order
            fb[(g << logW) + pos] = make_float2(accr, acci);
        }
        QD_STAMP_AT(2);
        // Park this tile's window(s): remember where their output goes, advance to the next tile.  The FFT + epilogue run
        // when the batch is full (or the workgroup has no further tile); the barrier below also keeps the next tile's
        // phase 1 from overwriting raw samples another wave's FIR is still reading.
        if constexpr (kBatch > 1) {
            if (tid == 0) {
                const uint64_t wr = w0 - P.out_window0;
                bmeta[4 * bslot + 0] = (uint32_t)wr; bmeta[4 * bslot + 1] = (uint32_t)(wr >> 32); bmeta[4 * bslot + 2] = g_cnt;
            }
        }
        ++bslot;
        if (dyn && tid == 0) {
            const uint64_t t2 = claim_resolve(claim);
            wq[0] = (uint32_t)t2; wq[1] = (uint32_t)(t2 >> 32);
        }
        const TileGeo tg_next = tile_geo<FMT, NT>(P, geo, tile_n, n_tiles);
        const uint32_t batch_lim = (kBatch > 1 && P.epi != 2 && !cf32_out) ? kBatch : 1u;   // the bucket / write sinks flush every tile
        const bool flush = bslot >= batch_lim || !tg_next.valid;                             // wave-uniform
        __syncthreads();
        tile = tile_n;
        if (dyn) tile_n = ((uint64_t)__builtin_amdgcn_readfirstlane(wq[1]) << 32) | __builtin_amdgcn_readfirstlane(wq[0]);
        else { walk_local += walk_step; tile_n = walk_tile(walk_local + walk_step); }
        QD_STAMP_AT(3);
        if (!flush) { rt_touch += rt_pf; tg = tg_next; QD_STAMP_TILE(); continue; }
        const uint32_t n_slots = bslot;
        bslot = 0;
        fb = fb0;

        __builtin_amdgcn_s_setprio(3);
        // ---------------- phase 3: FFT (rustfft Radix4: base butterflies, then radix-4 layers) over the batch
        const uint32_t fft_windows = kBatch > 1 ? n_slots * geo.G : g_cnt;      // parked slots hold G windows each (a short last tile leaves stale ones: transformed, never stored)
        if (!QD_DBG(P, 4) && !cf32_out) {
            const uint32_t base = geo.base_len;
            const uint32_t log_tpw = logW - geo.log_base;   // base tasks per window = W / base
            const uint32_t n_task = fft_windows << log_tpw;
            for (uint32_t t = tid; t < n_task; t += NT) {
                // windows are contiguous in fb, so task t owns chunk t
                float2 *d = fb + (size_t)t * base;
                if (base == 16) {
                    float2 v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = d[i];
                    bf16(v, P.tw16_1, P.tw16_2, P.tw16_3, P.root2);
#pragma unroll
                    for (int i = 0; i < 16; ++i) d[i] = v[i];
                } else if (base == 8) {
                    float2 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = d[i];
                    bf8(v, P.root2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] = v[i];
                } else if (base == 4) {
                    float2 v0 = d[0], v1 = d[1], v2 = d[2], v3 = d[3];
                    bf4(v0, v1, v2, v3);
                    d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
                } else if (base == 2) {
                    float2 v0 = d[0], v1 = d[1];
                    bf2(v0, v1);
                    d[0] = v0; d[1] = v1;
                }
            }
            uint32_t cols = base, log_cols = geo.log_base;
            const float2 *tw = twl;
            for (uint32_t layer = 0; layer < geo.layers; ++layer) {
                __syncthreads();
                const uint32_t n_bf = (fft_windows << logW) >> 2;   // W/4 butterflies per window
                for (uint32_t t = tid; t < n_bf; t += NT) {
                    // butterfly t: chunk (of 4*cols; windows are contiguous) t >> log_cols, column t & (cols-1)
                    const uint32_t chunk = t >> log_cols, i = t & (cols - 1);
                    float2 *d = fb + (size_t)chunk * 4 * cols + i;
                    float2 s0 = d[0];
                    float2 s1 = cmul(d[cols], tw[3 * i]);
                    float2 s2 = cmul(d[2 * cols], tw[3 * i + 1]);
                    float2 s3 = cmul(d[3 * cols], tw[3 * i + 2]);
                    bf4(s0, s1, s2, s3);
                    d[0] = s0; d[cols] = s1; d[2 * cols] = s2; d[3 * cols] = s3;
                }
                tw += 3 * cols;
                cols *= 4;
                log_cols += 2;
            }
        }
        QD_STAMP_AT(4);
        __syncthreads();
        QD_STAMP_AT(5);

        // ---------------- phase 4: fftshift + norm + epilogue (out index is tile base + o: coalesced), slot by slot
        for (uint32_t sl = 0; sl < n_slots; ++sl) {
        uint64_t wrel = w0 - P.out_window0;
        uint32_t n_out_s = n_out;
        const float2 *fbs = fb;
        if constexpr (kBatch > 1) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane(bmeta[4 * sl + 0]), hi = __builtin_amdgcn_readfirstlane(bmeta[4 * sl + 1]);
            wrel = ((uint64_t)hi << 32) | lo;
            n_out_s = __builtin_amdgcn_readfirstlane(bmeta[4 * sl + 2]) << logW;
            fbs = fb + (size_t)sl * geo.G * geo.W;
        }
        if (cf32_out) {
            // do_write / LowPass::read_at output (src/lib.rs:206-209): the decimated cf32 samples themselves
            float2 *outc = reinterpret_cast<float2 *>(P.out) + (wrel << logW);
            for (uint32_t o = tid; o < n_out_s; o += NT) outc[o] = fbs[o];
        } else if (P.epi == 2) {
            // freq_levels (src/fft.rs:95-97): sequential f32 sums of |X[k]| over each half
            float *nb = reinterpret_cast<float *>(raw);       // raw tile is dead now
            for (uint32_t o = tid; o < n_out_s; o += NT) nb[o] = norm_ref(fbs[o]);
            __syncthreads();
            if (tid < g_cnt) {
                const float *p = nb + (tid << logW);
                float first = 0.f, second = 0.f;
                for (uint32_t k = 0; k < W / 2; ++k) first = first + p[k];
                for (uint32_t k = W / 2; k < W; ++k) second = second + p[k];
                uint32_t tt = tid;
                asm volatile("" : "+v"(tt));       // opaque lane offset: no hoisted (and spilled) per-lane output pointer, see below
                reinterpret_cast<uint8_t *>(P.out)[wrel + tt] = first < second ? 0 : 1;
            }
        } else {
            float *outf = reinterpret_cast<float *>(P.out) + (wrel << logW);     // uniform base
            uint8_t *outb = reinterpret_cast<uint8_t *>(P.out) + (wrel << logW);
            for (uint32_t o = tid; o < n_out_s; o += NT) {
                // Lane offsets are kept opaque so hipcc does not hoist per-lane addresses (tid-derived, loop invariant) out of
                // the tile loop into VGPRs that end up spilled: a scratch reload here waits with s_waitcnt vmcnt(0) — a full
                // drain of the next tile's prefetch loads in every epilogue.
                uint32_t oi = o;
                asm volatile("" : "+v"(oi));
                const float2 xv = fbs[oi ^ (W >> 1)];         // fftshift: bin (b + W/2) mod W of the same window
                const float nm = QD_DBG(P, 8) ? xv.x : norm_ref(xv);
                // the store address is uniform base + 32-bit lane offset, opaque for the same reason
                uint32_t oo = o;
                asm volatile("" : "+v"(oo));
                if (QD_DBG(P, 16)) { asm volatile("" :: "v"(nm)); continue; }      // timing-only ablation: no output store
                if (P.epi == 0) outf[oo] = nm;
                else outb[oo] = glyph_of<GeoT>(P, nm);
            }
        }
        }
        rt_touch += rt_pf;       // first use of the touch loads: a whole tile after they were issued
        QD_STAMP_AT(6);
        // No barrier here for the norm / glyph / cf32 epilogues: they only read fb, the next tile's phase 1 only
        // writes the raw region, and barrier 1 of the next tile orders everything before fb (or dec/trc) is
        // written again.  The bucket epilogue parks its norms IN the raw region, so it keeps the barrier.
        if (P.epi == 2) __syncthreads();
        QD_STAMP_AT(7);
        QD_STAMP_TILE();
        tg = tg_next;
    }
    if constexpr (kDefer) {
        constexpr uint32_t GW = GeoT::G_ct * GeoT::W_ct, GWP = kHalf ? GeoT::kHalfOut : GW, FL = GeoT::kPackedTile ? GWP / 2 : GWP;
        if (kQuadFft && P.epi != 2) {
            if (dprev_valid && (tid >> 6) >= FL / 64 && (tid >> 6) < FL / 64 + 4) quad_fft_epilogue(fb0 + (size_t)(dpar ^ 1u) * GW, dprev_w0, (tid >> 6) - FL / 64);
        } else
        if (dprev_valid && (tid >> 6) == FL / 64) wave_fft_epilogue(fb0 + (size_t)(dpar ^ 1u) * GW, dprev_w0, dprev_gcnt);
    }
    if (dyn && tid == 0) {       // the last workgroup to leave re-arms the queue for the next launch
        if (atomicAdd(&P.work[16 * 8], 1ull) == (unsigned long long)gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x <= 8; ++x) P.work[16 * x] = 0;
        }
    }
    QD_STAMP_FLUSH();
#ifdef QD_WGTIME
    if (P.stamps && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *w = P.stamps + 256 + 4ull * blockIdx.x;
        w[0] = wg_t0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = xcc & 0xf; w[3] = 1;
    }
#endif
    if (P.dbg == 0xdeadbeefu) reinterpret_cast<double *>(P.out)[tid] = rt_touch;   // never true: keeps rt_touch live
}

// ---------------------------------------------------------------- the role-split kernel (FixedGeo FLAGS_ bit 9)
//
// k_chain keeps a tile's phases in sequence on ALL waves of the workgroup: phase 1 (4 waves), barrier, FIR (2 waves), barrier,
// with the FFT of the previous tile on a third wave — a chain of ~11 000 cycles per 128-point window on the north_star shape,
// during which the vector units are 76 % busy (profiles/r02/cfg3p_summary.json).  The chain, not HBM, sets the rate there: the
// same launch shape streams 17 % faster with the arithmetic ablated.
//
// Here the roles are split across waves and run CONCURRENTLY on one tile buffer: four producer waves (unpack -> NCO -> LDS,
// with the next tile's rows prefetched in registers, exactly k_chain's row-aligned phase 1) and one consumer wave (FIR of the
// window's two halves, then FFT + |X| + store), 320 threads per workgroup, five waves per SIMD.  A window's outputs split in
// two halves of W/2; half 0 reads tile rows [0, RA], half 1 rows [RB, RCH) (RB <= RA + 1), so one step of the consumer always
// leaves a set of rows nobody reads, and the producers refill exactly those:
//     step 1   consumer: FIR of half 0 (rows 0 .. RA)         producers: rows RA+1 .. RCH-1 of THIS window
//     step 2   consumer: FIR of half 1 (rows RB .. RCH-1)     producers: rows 0 .. RB-1 of the NEXT window
//     step 3   consumer: FFT + epilogue (no raw rows)         producers: rows RB .. RA of the next window
// One s_barrier ends each step (every wave executes the same number of barriers: no flags, no polling, nothing to deadlock).
// The products, their order and every rounding are k_chain's: the same process_row, fir_pair and wave_fft_epilogue_fn.
constexpr uint32_t kGeoPipe = 512, kGeoPipeFftWave = 1024;
constexpr uint32_t kGeoLoadSc0 = 2048, kGeoLoadSc1 = 4096;   // development: cache-policy bits of the phase-1 stream loads (with kGeoNtLoads)
      // bit 10: a sixth wave takes the FFT + epilogue (384 threads, two FFT slots)
constexpr int kPipeThreads = 320;
template <int V> struct IntC { static constexpr int value = V; };

template <int FMT, class GeoT>
constexpr bool pipe_geometry_ok(int rch) {
    if constexpr (!GeoT::kFixed) return false;
    else {
        constexpr uint32_t ROW = 256u * FmtTraits<FMT>::SPL;
        constexpr uint32_t tile_raw = GeoT::W * GeoT::D + GeoT::T;
        return GeoT::G == 1 && GeoT::S == GeoT::W && GeoT::kPairFir && GeoT::W == 128 && (GeoT::S * GeoT::D) % ROW == 0 &&
               (uint32_t)rch == (tile_raw + ROW - 1) / ROW && GeoT::D % FmtTraits<FMT>::SPL == 0 && GeoT::T > GeoT::D;
    }
}

template <int FMT, int NCO, class GeoT, int RCH, int LB, int NT = kPipeThreads>
__global__ __launch_bounds__(NT, LB) void k_chain_pipe(const ChainParams P) {
    static_assert(NT == 320 || NT == 384, "four producer waves + the FIR wave (+ an FFT wave)");
    constexpr bool kFftWave = NT == 384;          // a sixth wave transforms the PREVIOUS window while the FIR wave filters this one (two FFT slots)
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    constexpr int SPL = FT::SPL;
    constexpr bool HAS_SHIFT = NCO != 0;
    static_assert(pipe_geometry_ok<FMT, GeoT>(RCH), "role-split kernel: non-overlapping 128-point windows on a row-aligned, 16-byte-row tile");
    constexpr uint32_t PT = 256;                                                   // producer threads
    constexpr uint32_t W = GeoT::W, D = GeoT::D, T = GeoT::T, Dp = GeoT::Dp, HO = W / 2;
    constexpr uint32_t ROW = PT * SPL, ROWB = ROW * FT::BPS, VECB = SPL * FT::BPS;
    constexpr uint32_t kTileRaw = W * D + T, kRem = kTileRaw - (RCH - 1) * ROW;
    constexpr int RA = (int)((GeoT::c + (HO - 1) * D + T - 1) / ROW);              // last row half 0 reads
    constexpr int RB = (int)((HO * D + GeoT::c) / ROW);                            // first row half 1 reads
    static_assert(RB <= RA + 1 && RA < RCH && RB >= 1, "half split");
    const GeoT geo(P);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *raw = reinterpret_cast<float2 *>(smem);                                // same layout as k_chain (host: lds_for)
    float2 *fb0 = raw + geo.lds_raw_elems;
    float2 *twl = fb0 + (size_t)GeoT::kBatch * W;
    float *tapl = reinterpret_cast<float *>(twl + W);
    float *lut = tapl + ((T + 3) & ~3u);
    constexpr uint32_t kLutElems = (FMT == 1 || FMT == 2) ? 256u : 0u;
    uint32_t *wq = reinterpret_cast<uint32_t *>(lut + kLutElems) + 4 * GeoT::kBatch;

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    {
        const uint32_t n_tw = W - geo.base_len;
        for (uint32_t i = tid; i < n_tw; i += NT) twl[i] = P.tw[i];
        for (uint32_t i = tid; i < T; i += NT) tapl[i] = P.taps[i];
    }
    __syncthreads();

    const uint64_t n_tiles = P.n_windows;
    // tile walk and dynamic tile queue: as in k_chain
    uint64_t walk_base = 0, walk_local = blockIdx.x, walk_step = gridDim.x, walk_limit = n_tiles;
    const bool xcd_walk = (gridDim.x & 7u) == 0 && n_tiles >= 8;
    const uint64_t n8 = (n_tiles + 7) / 8;
    auto xcd_limit = [&](uint32_t x) -> uint64_t { const uint64_t b = (uint64_t)x * n8; return b >= n_tiles ? 0 : (n_tiles - b < n8 ? n_tiles - b : n8); };
    if (xcd_walk) {
        walk_base = (uint64_t)(blockIdx.x & 7u) * n8;
        walk_local = blockIdx.x >> 3;
        walk_step = gridDim.x >> 3;
        walk_limit = xcd_limit(blockIdx.x & 7u);
    }
    auto walk_tile = [&](uint64_t local) -> uint64_t { return local < walk_limit ? walk_base + local : n_tiles; };
    const bool dyn = P.work != nullptr && xcd_walk;
    const uint32_t my_x = blockIdx.x & 7u;
    auto claim_resolve = [&](unsigned long long got) -> uint64_t {
        uint32_t mx = my_x;
        asm volatile("" : "+s"(mx));
        const uint64_t dyn_base = 2 * walk_step;
        if (got + dyn_base < xcd_limit(mx)) return (uint64_t)mx * n8 + dyn_base + got;
#pragma unroll 1
        for (uint32_t k = 1; k < 8; ++k) {                                // own eighth exhausted: help the others
            const uint32_t xx = (mx + k) & 7u;
            const uint64_t lim = xcd_limit(xx);
            if (lim <= dyn_base) continue;
            const unsigned long long i = atomicAdd(&P.work[16 * xx], 1ull);
            if (i + dyn_base < lim) return (uint64_t)xx * n8 + dyn_base + i;
        }
        return n_tiles;
    };
    uint64_t tile, tile_n;
    tile = walk_tile(walk_local);                                         // static for the first two rounds, dynamic or static after
    tile_n = walk_tile(walk_local + walk_step);

    // ---- producer state
    LaneRot lr[SPL];
    if (HAS_SHIFT && producer) {
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const uint32_t j = tid * SPL + u;
            const double2 cs = P.jtab[j];
            lr[u].jf = (double)j; lr[u].c = cs.x; lr[u].s = cs.y;
        }
    }
    const uint32_t lane_pad = pad_index(geo, tid * SPL);
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto n_start_of = [&](uint64_t t) -> uint64_t { return (P.first_window + t) * ((uint64_t)GeoT::S * D); };
    auto rsrc_of = [&](uint64_t t) {
        const uint64_t ns = n_start_of(t);
        const uint64_t left = (P.src_first + P.src_count - ns) * FT::BPS;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * FT::BPS, 0,
                                                 left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
    };
    Vec pf[RCH];
    auto load_row = [&](const decltype(rsrc_of(0)) &rsrc, int i) {
        constexpr uint32_t kLastVec = ((kRem * FT::BPS - 1) / VECB) * VECB;
        uint32_t voff = tid * VECB;
        if (i + 1 == RCH && kRem != ROW) voff = voff < kLastVec ? voff : kLastVec;
        constexpr int aux = ct_load_aux(GeoT::kFlags);
        if constexpr (sizeof(Vec) == 16) {
            const v4u_t w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(i * ROWB), aux);
            pf[i].x = w.x; pf[i].y = w.y; pf[i].z = w.z; pf[i].w = w.w;
        } else {
            const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(i * ROWB), aux);
            pf[i].x = w.x; pf[i].y = w.y;
        }
    };
    double rt_touch = 0.0;
    // rows [I0, I1) of tile t_cur: consume the prefetched row, park it, refill the slot with the same row of t_refill
    auto produce = [&](auto i0c, auto i1c, uint64_t t_cur, uint64_t t_refill) {
        constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
        if constexpr (I0 < I1) {
            const uint64_t ns = n_start_of(t_cur);
            const uint64_t t_pf = t_refill < n_tiles ? t_refill : t_cur;           // last tile of this workgroup: harmless re-loads
            const auto rsrc = rsrc_of(t_pf);
            const_f64_p rows = (const_f64_p)(uintptr_t)(P.rowtab + (ns / ROW - P.rowtab_row0));
            if constexpr (HAS_SHIFT && I0 == 0) {                                   // L2 touch of the NEXT tile's row bases (see prefetch_rowtab)
                uint32_t r = tid < (uint32_t)RCH ? tid : (uint32_t)RCH - 1;
                asm volatile("" : "+v"(r));
                rt_touch += P.rowtab[n_start_of(t_pf) / ROW - P.rowtab_row0 + r].c;
            }
            TileGeo gl{};
            gl.tile_raw = kTileRaw;
            RowBase rb_next{};
            if constexpr (HAS_SHIFT) rb_next = load_rowbase_at(rows, I0);
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const Vec v = pf[i];
                const RowBase rb = rb_next;
                if constexpr (HAS_SHIFT) { if (i + 1 < I1) rb_next = load_rowbase_at(rows, i + 1); }
                if (i + 1 < RCH || kRem == ROW) {
                    process_row<FMT, PT, NCO, true>(P, geo, gl, i * (int)ROW, tid, v, rb, lr, lane_pad, lut, raw);
                } else {
                    if (tid * SPL < kRem) process_row<FMT, PT, NCO, (kRem % SPL) == 0>(P, geo, gl, i * (int)ROW, tid, v, rb, lr, lane_pad, lut, raw);
                }
                __builtin_amdgcn_sched_barrier(0);       // refill slot i only after row i is consumed (see k_chain)
                load_row(rsrc, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // ---- consumer: one half of the window's FIR outputs (k = half * HO + lane), scattered for the FFT
    auto fir_half = [&](uint32_t half, float2 *fb) {
        uint32_t lane = tid & 63u;
        asm volatile("" : "+v"(lane));                   // per-lane LDS addresses are rebuilt per step, not hoisted and spilled
        const uint32_t k = half * HO + lane;
        uint32_t jmax = (W - k) * D + T / 2;
        if (jmax > T) jmax = T;
        const float2 *rp = raw + (size_t)(k + geo.a0) * Dp;
        const float2 v = fir_pair<GeoT>(rp, jmax, tapl);
        constexpr uint32_t log_width = 2 * GeoT::layers;
        const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
        fb[yy + (rev4(xx, geo.layers) << geo.log_base)] = v;
    };

    // Each role runs its OWN loop (same trip count, three barriers per trip): in one shared loop the producers' prefetch and
    // lane registers would be live across the consumer's FIR as well and the kernel would need the SUM of the two register
    // sets instead of the larger one (114 VGPRs against the 96 that five waves per SIMD allow).
    bool cur_valid = tile < n_tiles;
    QD_STAMP_DECL
    auto next_after = [&]() -> uint64_t {                                        // the tile after `tile_n`, read behind step 1's barrier
        if (dyn) return ((uint64_t)__builtin_amdgcn_readfirstlane(wq[1]) << 32) | __builtin_amdgcn_readfirstlane(wq[0]);
        walk_local += walk_step;
        return walk_tile(walk_local + walk_step);
    };
    if (producer) {
        __builtin_amdgcn_s_setprio(0);
        if (cur_valid) {
            const auto rsrc = rsrc_of(tile);
#pragma unroll
            for (int i = 0; i < RCH; ++i) load_row(rsrc, i);
            produce(IntC<0>{}, IntC<RB>{}, tile, tile_n);
            produce(IntC<RB>{}, IntC<RA + 1>{}, tile, tile_n);
        }
        // The claim for the tile after next is one atomic whose reply takes ~1-2 us under load: it is issued a whole step
        // before its reply is read (behind step 1's barrier it is already needed by every wave).
        unsigned long long claim = 0;
        if (dyn && tid == 0) claim = atomicAdd(&P.work[16 * my_x], 1ull);
        __syncthreads();
        QD_STAMP_START();
        while (cur_valid) {
            produce(IntC<RA + 1>{}, IntC<RCH>{}, tile, tile_n);                    // step 1
            if (dyn && tid == 0) {
                const uint64_t t2 = claim_resolve(claim);
                wq[0] = (uint32_t)t2; wq[1] = (uint32_t)(t2 >> 32);
            }
            QD_STAMP_AT(0);
            __syncthreads();
            QD_STAMP_AT(1);
            const uint64_t tile_nn = next_after();
            const bool next_valid = tile_n < n_tiles;
            if (dyn && tid == 0 && next_valid) claim = atomicAdd(&P.work[16 * my_x], 1ull);     // read at the end of the next step 1
            if (next_valid) produce(IntC<0>{}, IntC<RB>{}, tile_n, tile_nn);       // step 2
            QD_STAMP_AT(2);
            __syncthreads();
            QD_STAMP_AT(3);
            // step 3: the row(s) both halves read.  With the FFT on its own wave nothing else runs meanwhile, so go first.
            if constexpr (kFftWave) __builtin_amdgcn_s_setprio(3);
            if (next_valid) produce(IntC<RB>{}, IntC<RA + 1>{}, tile_n, tile_nn);
            if constexpr (kFftWave) __builtin_amdgcn_s_setprio(0);
            QD_STAMP_AT(4);
            __syncthreads();
            QD_STAMP_AT(5);
            QD_STAMP_TILE();
            tile = tile_n; tile_n = tile_nn; cur_valid = next_valid;
        }
    } else if (wave == 4) {
        __builtin_amdgcn_s_setprio(2);
        uint32_t par = 0;
        __syncthreads();
        QD_STAMP_START();
        while (cur_valid) {
            float2 *fb = fb0 + (kFftWave ? par * W : 0u);
            fir_half(0, fb);                                                       // step 1
            QD_STAMP_AT(0);
            __syncthreads();
            QD_STAMP_AT(1);
            const uint64_t tile_nn = next_after();
            const bool next_valid = tile_n < n_tiles;
            fir_half(1, fb);                                                       // step 2
            QD_STAMP_AT(2);
            __syncthreads();
            QD_STAMP_AT(3);
            if constexpr (!kFftWave) {                                             // step 3
                __builtin_amdgcn_s_setprio(3);
                wave_fft_epilogue_fn<GeoT>(P, geo, twl, fb, P.first_window + tile, 1u, tid);
                __builtin_amdgcn_s_setprio(2);
            }
            QD_STAMP_AT(4);
            __syncthreads();
            QD_STAMP_AT(5);
            par ^= 1u;
            tile = tile_n; tile_n = tile_nn; cur_valid = next_valid;
        }
    } else {
        // the FFT wave: window i-1 (parked in the other slot) during step 1 of window i
        __builtin_amdgcn_s_setprio(1);
        uint32_t par = 0;
        uint64_t prev = n_tiles;
        __syncthreads();
        QD_STAMP_START();
        while (cur_valid) {
            if (prev < n_tiles) wave_fft_epilogue_fn<GeoT>(P, geo, twl, fb0 + (par ^ 1u) * W, P.first_window + prev, 1u, tid);
            QD_STAMP_AT(0);
            __syncthreads();
            QD_STAMP_AT(1);
            const uint64_t tile_nn = next_after();
            const bool next_valid = tile_n < n_tiles;
            __syncthreads();
            QD_STAMP_AT(3);
            __syncthreads();
            QD_STAMP_AT(5);
            par ^= 1u;
            prev = tile;
            tile = tile_n; tile_n = tile_nn; cur_valid = next_valid;
        }
        if (prev < n_tiles) wave_fft_epilogue_fn<GeoT>(P, geo, twl, fb0 + (par ^ 1u) * W, P.first_window + prev, 1u, tid);
    }
    QD_STAMP_FLUSH();
    if (dyn && tid == 0) {       // the last workgroup to leave re-arms the queue for the next launch
        if (atomicAdd(&P.work[16 * 8], 1ull) == (unsigned long long)gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x <= 8; ++x) P.work[16 * x] = 0;
        }
    }
    if (P.dbg == 0xdeadbeefu) reinterpret_cast<double *>(P.out)[tid] = rt_touch;   // never true: keeps rt_touch live
}

// ---------------------------------------------------------------- the three-stage kernel for overlapping windows (FixedGeo FLAGS_ bit 15)
//
// Overlapping-window chains with long filters (cfg3, the README's FSK chain: 400 taps, W = 64, S = 16) run k_chain with ONE
// ~156 KiB tile of 27 windows per CU, so a tile's phases — unpack + NCO, the shared FIR, gather + FFT, |X| — run strictly one
// after the other and add up (13 k + 12.4 k + 6.4 k stamped cycles): the vector units idle while the FIR
// waits on LDS, the LDS idles while the NCO computes.  Unlike the 128-point shapes (k_chain_pipe), here a tile's FIR is ONE pass
// over all of the tile's outputs, and two half-size tiles fit in LDS: so the three stages run CONCURRENTLY on consecutive tiles,
//     waves 0-7   (512 threads)  phase 1 of tile i+1 into raw[(i+1)&1]   (next tile's rows prefetched in registers)
//     waves 8-11  (256 lanes)    shared FIR of tile i from raw[i&1] into dec / trc[i&1]   (fir_pair, packed, taps from LDS)
//     waves 12-15 (4 waves)      tile i-1: gather from dec / trc[(i-1)&1], FFT, |X|, store — each wave its own windows, wave-local
// with one s_barrier per tile (every wave executes the same number of barriers: nothing to deadlock).  Same products, order and
// roundings as k_chain's shared-FIR path; the price is the halo of a 12-window tile (31 % of its samples against 14 %).
constexpr uint32_t kGeoPipe3 = 32768;
constexpr int kPipe3Threads = 1024, kPipe3Prod = 512;

template <int FMT, class GeoT>
constexpr bool pipe3_geometry_ok(int rch) {
    if constexpr (!GeoT::kFixed) return false;
    else {
        constexpr uint32_t ROW = (uint32_t)kPipe3Prod * FmtTraits<FMT>::SPL;
        constexpr uint32_t tile_raw = (GeoT::G - 1) * GeoT::S * GeoT::D + GeoT::W * GeoT::D + GeoT::T;
        constexpr uint32_t Q = (GeoT::G - 1) * GeoT::S + GeoT::W;
        return GeoT::kShared && GeoT::kUnrolledShared && Q <= 256 && (GeoT::G * GeoT::S * GeoT::D) % ROW == 0 &&
               (uint32_t)rch == (tile_raw + ROW - 1) / ROW && GeoT::D % FmtTraits<FMT>::SPL == 0 && GeoT::W <= 64 * 16 && GeoT::G >= 1;
    }
}
// LDS of the three-stage kernel, in float2 elements before the taps (the host restates this: lds_for_pipe3)
template <class GeoT> constexpr uint32_t pipe3_q_pad() { return (((GeoT::G - 1) * GeoT::S + GeoT::W) + 1) & ~1u; }

template <int FMT, int NCO, class GeoT, int RCH, int LB>
__global__ __launch_bounds__(kPipe3Threads, LB) void k_chain_pipe3(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    constexpr int SPL = FT::SPL;
    constexpr bool HAS_SHIFT = NCO != 0;
    static_assert(pipe3_geometry_ok<FMT, GeoT>(RCH), "three-stage kernel: overlapping windows, straight-line shared FIR, <= 256 outputs per tile, row-aligned tiles");
    constexpr uint32_t PT = kPipe3Prod;
    constexpr uint32_t W = GeoT::W, S = GeoT::S, D = GeoT::D, T = GeoT::T, G = GeoT::G, Dp = GeoT::Dp, logW = GeoT::logW;
    constexpr uint32_t ROW = PT * SPL, ROWB = ROW * FT::BPS, VECB = SPL * FT::BPS;
    constexpr uint32_t kTileRaw = (G - 1) * S * D + W * D + T, kRem = kTileRaw - (RCH - 1) * ROW;
    constexpr uint32_t Q = (G - 1) * S + W, QP = pipe3_q_pad<GeoT>();
    constexpr uint32_t GV = (G + 3) / 4;                                          // windows per FFT wave
    const GeoT geo(P);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *raw0 = reinterpret_cast<float2 *>(smem);
    float2 *dec0 = raw0 + 2 * (size_t)geo.lds_raw_elems;                           // dec / trc of set 0, then of set 1
    float2 *fbx = dec0 + 4 * (size_t)QP;                                           // G*W: the FFT waves' buffers (wave v: windows [v GV, (v+1) GV))
    float2 *twl = fbx + (size_t)G * W;
    float *tapl = reinterpret_cast<float *>(twl + W);
    uint32_t *wq = reinterpret_cast<uint32_t *>(tapl + ((T + 3) & ~3u));

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const uint32_t n_tw = W - geo.base_len;
        for (uint32_t i = tid; i < n_tw; i += kPipe3Threads) twl[i] = P.tw[i];
        for (uint32_t i = tid; i < T; i += kPipe3Threads) tapl[i] = P.taps[i];
    }
    __syncthreads();

    const uint64_t n_tiles = (P.n_windows + G - 1) / G;
    uint64_t walk_base = 0, walk_local = blockIdx.x, walk_step = gridDim.x, walk_limit = n_tiles;
    const bool xcd_walk = (gridDim.x & 7u) == 0 && n_tiles >= 8;
    const uint64_t n8 = (n_tiles + 7) / 8;
    auto xcd_limit = [&](uint32_t x) -> uint64_t { const uint64_t b = (uint64_t)x * n8; return b >= n_tiles ? 0 : (n_tiles - b < n8 ? n_tiles - b : n8); };
    if (xcd_walk) {
        walk_base = (uint64_t)(blockIdx.x & 7u) * n8;
        walk_local = blockIdx.x >> 3;
        walk_step = gridDim.x >> 3;
        walk_limit = xcd_limit(blockIdx.x & 7u);
    }
    auto walk_tile = [&](uint64_t local) -> uint64_t { return local < walk_limit ? walk_base + local : n_tiles; };
    const bool dyn = P.work != nullptr && xcd_walk;
    const uint32_t my_x = blockIdx.x & 7u;
    auto claim_resolve = [&](unsigned long long got) -> uint64_t {                // three static rounds, then the counters
        uint32_t mx = my_x;
        asm volatile("" : "+s"(mx));
        const uint64_t dyn_base = 3 * walk_step;
        if (got + dyn_base < xcd_limit(mx)) return (uint64_t)mx * n8 + dyn_base + got;
#pragma unroll 1
        for (uint32_t k = 1; k < 8; ++k) {
            const uint32_t xx = (mx + k) & 7u;
            const uint64_t lim = xcd_limit(xx);
            if (lim <= dyn_base) continue;
            const unsigned long long i = atomicAdd(&P.work[16 * xx], 1ull);
            if (i + dyn_base < lim) return (uint64_t)xx * n8 + dyn_base + i;
        }
        return n_tiles;
    };
    // the pipeline's tiles: x (FFT stage), f (FIR stage), p (being produced), pn (prefetched next)
    uint64_t tile_x = n_tiles, tile_f = walk_tile(walk_local), tile_p = walk_tile(walk_local + walk_step), tile_pn = walk_tile(walk_local + 2 * walk_step);
    walk_local += 2 * walk_step;                                                   // static walk: walk_local names tile_pn's round
    auto next_after = [&]() -> uint64_t {                                          // the tile after tile_pn, read behind the barrier
        if (dyn) return ((uint64_t)__builtin_amdgcn_readfirstlane(wq[1]) << 32) | __builtin_amdgcn_readfirstlane(wq[0]);
        walk_local += walk_step;
        return walk_tile(walk_local);
    };
    auto g_cnt_of = [&](uint64_t t) -> uint32_t {
        const uint64_t w0 = t * G, left = P.n_windows - w0;
        return left < G ? (uint32_t)left : G;
    };

    if (wave < 8) {
        // ================= producers
        LaneRot lr[SPL];
        if constexpr (HAS_SHIFT) {
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const uint32_t j = tid * SPL + u;
                const double2 cs = P.jtab[j];
                lr[u].jf = (double)j; lr[u].c = cs.x; lr[u].s = cs.y;
            }
        }
        const uint32_t lane_pad = pad_index(geo, tid * SPL);
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
        auto n_start_of = [&](uint64_t t) -> uint64_t { return (P.first_window + t * G) * ((uint64_t)S * D); };
        auto rsrc_of = [&](uint64_t t) {
            const uint64_t ns = n_start_of(t);
            const uint64_t left = (P.src_first + P.src_count - ns) * FT::BPS;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * FT::BPS, 0,
                                                     left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
        };
        Vec pf[RCH];
        auto load_row = [&](const decltype(rsrc_of(0)) &rsrc, int i) {
            constexpr uint32_t kLastVec = ((kRem * FT::BPS - 1) / VECB) * VECB;
            uint32_t voff = tid * VECB;
            if (i + 1 == RCH && kRem != ROW) voff = voff < kLastVec ? voff : kLastVec;
            constexpr int aux = ct_load_aux(GeoT::kFlags);
            // bit 16: the tile's first and last row are the neighbouring tile's too and keep the default policy (see k_chain)
            const bool shared_row = (GeoT::kFlags & kGeoNtInner) && (i == 0 || i + 1 == RCH);
            if constexpr (sizeof(Vec) == 16) {
                const v4u_t w = shared_row ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(i * ROWB), 0)
                                           : __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(i * ROWB), aux);
                pf[i].x = w.x; pf[i].y = w.y; pf[i].z = w.z; pf[i].w = w.w;
            } else {
                const v2u_t w = shared_row ? __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(i * ROWB), 0)
                                           : __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(i * ROWB), aux);
                pf[i].x = w.x; pf[i].y = w.y;
            }
        };
        double rt_touch = 0.0;
        auto produce = [&](uint64_t t_cur, uint64_t t_refill, float2 *rawbuf) {
            const uint64_t ns = n_start_of(t_cur);
            const uint64_t t_pf = t_refill < n_tiles ? t_refill : t_cur;          // last tile of this workgroup: harmless re-loads
            const auto rsrc = rsrc_of(t_pf);
            const_f64_p rows = (const_f64_p)(uintptr_t)(P.rowtab + (ns / ROW - P.rowtab_row0));
            if constexpr (HAS_SHIFT) {                                             // L2 touch of the next tile's row bases
                uint32_t r = tid < (uint32_t)RCH ? tid : (uint32_t)RCH - 1;
                asm volatile("" : "+v"(r));
                rt_touch += P.rowtab[n_start_of(t_pf) / ROW - P.rowtab_row0 + r].c;
            }
            TileGeo gl{};
            gl.tile_raw = kTileRaw;
            RowBase rb_next{};
            if constexpr (HAS_SHIFT) rb_next = load_rowbase_at(rows, 0);
#pragma unroll
            for (int i = 0; i < RCH; ++i) {
                const Vec v = pf[i];
                const RowBase rb = rb_next;
                if constexpr (HAS_SHIFT) { if (i + 1 < RCH) rb_next = load_rowbase_at(rows, i + 1); }
                if (i + 1 < RCH || kRem == ROW) {
                    process_row<FMT, PT, NCO, true>(P, geo, gl, i * (int)ROW, tid, v, rb, lr, lane_pad, nullptr, rawbuf);
                } else {
                    if (tid * SPL < kRem) process_row<FMT, PT, NCO, (kRem % SPL) == 0>(P, geo, gl, i * (int)ROW, tid, v, rb, lr, lane_pad, nullptr, rawbuf);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_row(rsrc, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_s_setprio(0);
        uint32_t par = 0;
        if (tile_f < n_tiles) {
            const auto rsrc = rsrc_of(tile_f);
#pragma unroll
            for (int i = 0; i < RCH; ++i) load_row(rsrc, i);
            produce(tile_f, tile_p, raw0);
        }
        __syncthreads();
        while (tile_f < n_tiles || tile_x < n_tiles) {
            if (tile_p < n_tiles) produce(tile_p, tile_pn, raw0 + (size_t)(par ^ 1u) * geo.lds_raw_elems);
            __syncthreads();
            const uint64_t nx = next_after();
            tile_x = tile_f; tile_f = tile_p; tile_p = tile_pn; tile_pn = nx; par ^= 1u;
        }
        if (P.dbg == 0xdeadbeefu) reinterpret_cast<double *>(P.out)[tid] = rt_touch;   // never true: keeps rt_touch live
    } else if (wave < 12) {
        // ================= the shared FIR: every decimated output of the tile once, full chain + the truncated snapshot
        __builtin_amdgcn_s_setprio(2);
        uint32_t par = 0;
        constexpr uint32_t c_half = T - T / 2, ntrunc = c_half ? (c_half + D - 1) / D - 1 : 0;
        __syncthreads();
        while (tile_f < n_tiles || tile_x < n_tiles) {
            unsigned long long claim = 0;
            if (dyn && tid == (uint32_t)kPipe3Prod) claim = atomicAdd(&P.work[16 * my_x], 1ull);     // this wave issues no other vector memory
            if (tile_f < n_tiles) {
                const uint32_t g_cnt = g_cnt_of(tile_f);
                uint32_t qi = tid - (uint32_t)kPipe3Prod;
                asm volatile("" : "+v"(qi));
                if (qi < Q) {
                    uint32_t jmax = T;
                    if (qi + ntrunc >= W) {                      // may be in the truncated tail of window g
                        const uint32_t g = (qi - (W - ntrunc)) / S, k = qi - g * S;
                        if (k < W && g < g_cnt) { const uint32_t jm = (W - k) * D + T / 2; if (jm < T) jmax = jm; }
                    }
                    const float2 *rowp = raw0 + (size_t)par * geo.lds_raw_elems + (size_t)(qi + geo.a0) * Dp;
                    float2 snap = make_float2(0.f, 0.f);
                    const float2 full = fir_pair<GeoT, true>(rowp, jmax, tapl, &snap);
                    float2 *dec = dec0 + (size_t)par * 2 * QP, *trc = dec + QP;
                    dec[qi] = full;
                    if (jmax < T) trc[qi] = snap;
                }
            }
            if (dyn && tid == (uint32_t)kPipe3Prod) {
                const uint64_t t2 = claim_resolve(claim);
                wq[0] = (uint32_t)t2; wq[1] = (uint32_t)(t2 >> 32);
            }
            __syncthreads();
            const uint64_t nx = next_after();
            tile_x = tile_f; tile_f = tile_p; tile_p = tile_pn; tile_pn = nx; par ^= 1u;
        }
    } else {
        // ================= gather + FFT + |X| of the tile filtered one step earlier: wave v takes windows [v GV, (v+1) GV), all wave-local
        __builtin_amdgcn_s_setprio(1);
        uint32_t par = 0;
        const uint32_t v = wave - 12;
        __syncthreads();
        while (tile_f < n_tiles || tile_x < n_tiles) {
            if (tile_x < n_tiles) {
                const uint32_t g_cnt = g_cnt_of(tile_x);
                const uint32_t g0 = v * GV, g1 = (g0 + GV < g_cnt) ? g0 + GV : g_cnt;
                if (g0 < g1) {
                    const float2 *dec = dec0 + (size_t)(par ^ 1u) * 2 * QP, *trc = dec + QP;
                    float2 *fbw = fbx + (size_t)g0 * W;
                    uint32_t lane = tid & 63u;
                    asm volatile("" : "+v"(lane));
                    constexpr uint32_t log_width = 2 * GeoT::layers;
                    const uint32_t n_o = (g1 - g0) << logW;
                    for (uint32_t o = lane; o < n_o; o += 64) {
                        const uint32_t gl_ = o >> logW, k = o & (W - 1);
                        const uint32_t qi = (g0 + gl_) * S + k;
                        const bool tr = (W - k) * D + T / 2 < T;
                        const float2 val = tr ? trc[qi] : dec[qi];
                        const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                        fbw[(gl_ << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = val;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    wave_fft_epilogue_fn<GeoT>(P, geo, twl, fbw, P.first_window + tile_x * G + g0, g1 - g0, tid);
                }
            }
            __syncthreads();
            const uint64_t nx = next_after();
            tile_x = tile_f; tile_f = tile_p; tile_p = tile_pn; tile_pn = nx; par ^= 1u;
        }
    }
    if (dyn && tid == 0) {       // the last workgroup to leave re-arms the queue for the next launch
        if (atomicAdd(&P.work[16 * 8], 1ull) == (unsigned long long)gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x <= 8; ++x) P.work[16 * x] = 0;
        }
    }
}

// LDS swizzle of the wave's transform buffer (plan-time builds, widths 8 ... 64).  Unswizzled, the buffer's accesses conflict: the
// scatter into transposed order 2- to 4-way, the base butterflies' 16-byte pieces (a lane owns a run of `base` points: lane stride 64 or
// 128 B) 4- to 8-way on every read and write, the first radix-4 layer 2- to 4-way — rocprofv3 on cf32 W = 64 with a shift: the LDS array busy
// 84 % of the kernel's cycles, 59 % of them conflicts (profiles/r04/nofir_pmc_spark_v1.log).  sigma(p) = p ^ M p with M a GF(2) matrix
// that only feeds HIGHER index bits into bits 1 ... 4 (a bijection; bit 0 untouched, so 16-byte pieces stay whole; sources >= log2(base),
// so a run sees one XOR value).  m[d - 1] = the source bits XORed into bit d, found per (W, SPL) by scripts/lds_swizzle_search.py over
// every LDS instruction of a 512- and a 1024-sample tile under the bank rules of MI355X_MICROARCH.md (lane-group cycles, identity -> map;
// conflict-free = 1x):
template <uint32_t W, uint32_t SPL> struct SparkSwz { static constexpr uint32_t m[4] = {0, 0, 0, 0}; };
template <> struct SparkSwz<64, 2> { static constexpr uint32_t m[4] = {0x40, 0x90, 0x20, 0x40}; };   // 3.30x -> 0.90x (some groups idle)
template <> struct SparkSwz<32, 2> { static constexpr uint32_t m[4] = {0x8, 0x10, 0x30, 0x40}; };    // 2.78x -> 1.00x
template <> struct SparkSwz<16, 2> { static constexpr uint32_t m[4] = {0x10, 0x20, 0x40, 0x80}; };   // 4.14x -> 1.14x
template <> struct SparkSwz<8, 2> { static constexpr uint32_t m[4] = {0x50, 0x20, 0x0, 0x0}; };      // 2.83x -> 1.33x
template <> struct SparkSwz<64, 4> { static constexpr uint32_t m[4] = {0x40, 0x90, 0x20, 0x40}; };   // 3.10x -> 0.90x
template <> struct SparkSwz<32, 4> { static constexpr uint32_t m[4] = {0x8, 0x10, 0x20, 0x40}; };    // 2.78x -> 1.00x
template <> struct SparkSwz<16, 4> { static constexpr uint32_t m[4] = {0x10, 0x20, 0x40, 0x80}; };   // 4.71x -> 1.14x
template <> struct SparkSwz<8, 4> { static constexpr uint32_t m[4] = {0x50, 0x20, 0x0, 0x0}; };      // 3.50x -> 1.33x
// SPL = 1: the scatter is a gather of consecutive decimated outputs, one per lane (k_chain_pipe3s's transform stage)
template <> struct SparkSwz<64, 1> { static constexpr uint32_t m[4] = {0x40, 0x90, 0x20, 0x40}; };   // 3.70x -> 0.90x
template <> struct SparkSwz<128, 1> { static constexpr uint32_t m[4] = {0x8, 0x10, 0x20, 0x40}; };   // 3.33x -> 1.17x
template <> struct SparkSwz<32, 1> { static constexpr uint32_t m[4] = {0x8, 0x10, 0x20, 0x40}; };    // 2.78x -> 1.00x
template <> struct SparkSwz<16, 1> { static constexpr uint32_t m[4] = {0x10, 0x20, 0x40, 0x80}; };   // 3.86x -> 0.86x
template <> struct SparkSwz<256, 1> { static constexpr uint32_t m[4] = {0x40, 0x90, 0x20, 0x40}; };  // 4.92x -> 1.08x
template <class SZ> struct SparkSwzFn {
    // the sources at distance `off` above their destination, as a mask over the destination bits
    static constexpr uint32_t mask_off(uint32_t off) {
        uint32_t r = 0;
        for (uint32_t d = 1; d <= 4; ++d) if ((SZ::m[d - 1] >> (d + off)) & 1u) r |= 1u << d;
        return r;
    }
    // sigma(p) ^ p.  GF(2)-linear: delta(a | b) = delta(a) ^ delta(b) for disjoint a, b — callers split an index into a per-lane part
    // (formed once per tile) and a compile-time part (folded)
    static __device__ __forceinline__ constexpr uint32_t delta(uint32_t p) {
        constexpr uint32_t k1 = mask_off(1), k2 = mask_off(2), k3 = mask_off(3), k4 = mask_off(4), k5 = mask_off(5), k6 = mask_off(6), k7 = mask_off(7), k8 = mask_off(8);
        return ((p >> 1) & k1) ^ ((p >> 2) & k2) ^ ((p >> 3) & k3) ^ ((p >> 4) & k4) ^ ((p >> 5) & k5) ^ ((p >> 6) & k6) ^ ((p >> 7) & k7) ^ ((p >> 8) & k8);
    }
};

// The transform of a wave's tile in the swizzled layout: base butterflies on 16-byte pieces, radix-4 layers in place (W >= 8, compile-time
// geometry; the arithmetic and its order are wave_fft_epilogue_fn's).  Addresses are LDS BYTE offsets: the buffer starts on a 256-byte
// boundary (k_spark checks), so the swizzle's XORs (index bits 1 ... 4 = byte bits 4 ... 7) apply to the byte address directly — one v_xor
// per access on top of a per-lane base, instead of an XOR, a shift and an add.
typedef float spark_f2n __attribute__((ext_vector_type(2)));
typedef float spark_f4n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) spark_f2n spark_lds_f2;
typedef __attribute__((address_space(3))) spark_f4n spark_lds_f4;
__device__ __forceinline__ float2 spark_ld2(uint32_t a) { const spark_f2n v = *(const spark_lds_f2 *)(uintptr_t)a; return make_float2(v.x, v.y); }
__device__ __forceinline__ void spark_st2(uint32_t a, float2 v) { *(spark_lds_f2 *)(uintptr_t)a = spark_f2n{v.x, v.y}; }

template <class GeoT, uint32_t TS, uint32_t SPL, int PART = 3 /* 1: base butterflies; 2: radix-4 layers; 3: both */, bool RT = false /* n_win windows of the buffer hold data (else all TS / W) */>
__device__ __forceinline__ void spark_fft_swz(const ChainParams &P, const float2 *twl, uint32_t fb /* LDS byte offset of the wave's buffer */, uint32_t lane_in,
                                              uint32_t n_win = TS / GeoT::W) {
    using SZ = SparkSwzFn<SparkSwz<GeoT::W, SPL>>;
    constexpr uint32_t base = GeoT::base_len, lb = GeoT::log_base, layers = GeoT::layers;
    static_assert(base == 8 || base == 16, "spark_fft_swz: W >= 8");
    uint32_t lo = lane_in;
    asm volatile("" : "+v"(lo));            // opaque per tile: addresses are rebuilt, not hoisted out of the tile loop and spilled
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    constexpr uint32_t n_task = TS / base;
    const uint32_t n_task_rt = n_win << (GeoT::logW - lb);
    if constexpr ((PART & 1) != 0) {
#pragma unroll
    for (uint32_t k = 0; k < (n_task + 63) / 64; ++k) {
        const uint32_t t = lo + 64 * k;
        if (RT ? t < n_task_rt : (n_task % 64 == 0 || t < n_task)) {
            const uint32_t r0 = t << lb, a0 = fb + ((r0 ^ SZ::delta(r0)) << 3);          // piece q of the run sits at a0 ^ 16 q
            float2 v[base];
#pragma unroll
            for (uint32_t q = 0; q < base / 2; ++q) {
                const spark_f4n pc = *(const spark_lds_f4 *)(uintptr_t)(a0 ^ (16 * q));
                v[2 * q] = make_float2(pc.x, pc.y); v[2 * q + 1] = make_float2(pc.z, pc.w);
            }
            if constexpr (base == 16) bf16(v, P.tw16_1, P.tw16_2, P.tw16_3, P.root2); else bf8(v, P.root2);
            uint32_t a1 = a0;
            asm volatile("" : "+v"(a1));        // the piece addresses are formed again (one XOR each) instead of held across the butterfly
#pragma unroll
            for (uint32_t q = 0; q < base / 2; ++q)
                *(spark_lds_f4 *)(uintptr_t)(a1 ^ (16 * q)) = spark_f4n{v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y};
        }
    }
    }
    uint32_t cols = base, log_cols = lb, tw_off = 0;
    const uint32_t n_bf_rt = n_win << (GeoT::logW - 2);
    if constexpr ((PART & 2) != 0) {
#pragma unroll
    for (uint32_t l = 0; l < layers; ++l) {
        wsync();
        // layers of at most 64 columns: a lane's butterflies t = lane + 64 k share i = t mod cols, hence their three twiddles
        float2 tc1 = make_float2(0.f, 0.f), tc2 = tc1, tc3 = tc1;
        if (cols <= 64) { const uint32_t i = lo & (cols - 1); tc1 = twl[tw_off + 3 * i]; tc2 = twl[tw_off + 3 * i + 1]; tc3 = twl[tw_off + 3 * i + 2]; }
#pragma unroll
        for (uint32_t k = 0; k < (TS / 4 + 63) / 64; ++k) {
            const uint32_t t = lo + 64 * k, chunk = t >> log_cols, i = t & (cols - 1);
            if (RT ? t >= n_bf_rt : ((TS / 4) % 64 != 0 && t >= TS / 4)) continue;
            const uint32_t p0 = chunk * 4 * cols + i, b0 = fb + ((p0 ^ SZ::delta(p0)) << 3);
            uint32_t dp[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                // q cols: index bits log_cols, log_cols + 1, zero in p0.  Bits below 5 (and the XOR value, bits 1 ... 4) go in by XOR — inside the
                // low 256 bytes, where the 256-byte-aligned base contributes nothing —, bits from 5 up by addition (an instruction offset)
                const uint32_t qc = q * cols, lo_q = (qc ^ SZ::delta(qc)) & 31u, hi_q = qc & ~31u;
                dp[q] = (b0 ^ (lo_q << 3)) + (hi_q << 3);
            }
            float2 t1 = tc1, t2 = tc2, t3 = tc3;
            if (cols > 64) { t1 = twl[tw_off + 3 * i]; t2 = twl[tw_off + 3 * i + 1]; t3 = twl[tw_off + 3 * i + 2]; }
            float2 s0 = spark_ld2(dp[0]);
            float2 s1 = cmul(spark_ld2(dp[1]), t1);
            float2 s2 = cmul(spark_ld2(dp[2]), t2);
            float2 s3 = cmul(spark_ld2(dp[3]), t3);
            bf4(s0, s1, s2, s3);
            spark_st2(dp[0], s0); spark_st2(dp[1], s1); spark_st2(dp[2], s2); spark_st2(dp[3], s3);
        }
        tw_off += 3 * cols; cols *= 4; log_cols += 2;
    }
    }
    wsync();
}

// ---------------------------------------------------------------- the STREAMING three-stage kernel (FixedGeo FLAGS_ bits 15 + 17)
//
// k_chain_pipe3 treats tiles as independent: every tile re-reads, re-shifts and re-filters the (W - S) D + T samples it shares with
// its predecessor — for a 12-window tile of the 64-point / stride-16 chains 31 % of phase 1 and 25 % of the shared FIR are work the
// previous tile has already done.  Here a workgroup owns a CONTIGUOUS run of tiles and carries that state in LDS:
//   * the shifted samples live in a ring of RR rows (+ a mirror of the ring's first (b0 + T) samples behind its end, so that a
//     chain's straight-line reads never wrap); a step adds exactly N = G S D new samples (RN rows),
//   * the decimated outputs live in a ring of 3 G S entries (dec) + their truncated snapshots (trc); a step adds exactly G S new
//     full outputs — the ones whose last tap became available with the step's rows — on G S lanes,
//   * the FFT stage gathers its windows out of the output ring.
// Step s = -1 of a run is the cold start (rows [0, N) of the run, outputs [0, f0)), steps 0 .. n-1 are its tiles; the stages run
// one step apart, one barrier per step, as in k_chain_pipe3.  Runs are a static, equal split of the launch's tiles (no queue: a
// claimed tile would have to be its predecessor's neighbour).  Products, their order and every rounding are unchanged: output q
// is the same fir_pair chain over the same shifted samples, computed once instead of up to twice.
constexpr uint32_t kGeoStream = 131072;
// bit 18: the streaming kernel as the `write` sink (QD_EPI_CF32_BLOCKS, src/lib.rs:178-213): producers + FIR waves only; the FIR lanes store
// their decimated outputs themselves (a wave's 64 outputs are 512 contiguous bytes), truncation relative to the read_at block (ChainParams::blk_len)
constexpr uint32_t kGeoWriteSink = 262144;

template <int FMT, class GeoT, int PT_ = kPipe3Prod>
struct Pipe3S {
    static_assert(PT_ == 256 || PT_ == 512, "producer threads: 256 or 512");
    static constexpr uint32_t SPL = FmtTraits<FMT>::SPL, ROW = (uint32_t)PT_ * SPL;      // PT_ producer threads (PT_ / 64 waves), then four FIR and four FFT waves
    static constexpr uint32_t W = GeoT::W, S = GeoT::S, D = GeoT::D, T = GeoT::T, G = GeoT::G, Dp = GeoT::Dp;
    static constexpr uint32_t N = G * S * D, RN = N / ROW, GS = G * S;
    static constexpr uint32_t c_half = T - T / 2, ntrunc = c_half ? (c_half + D - 1) / D - 1 : 0;
    static constexpr uint32_t f0 = N >= c_half + T ? (N - c_half - T) / D + 1 : 0;          // full outputs the cold start's rows complete
    static constexpr uint32_t RR = (2 * N + T + 2 * D + ROW - 1) / ROW;                       // ring rows: two steps + a chain's look-back
    static constexpr uint32_t RINGD = RR * (ROW / D);                                           // ... in LDS rows of D samples
    static constexpr uint32_t MIRD = (GeoT::b0 + T + D - 1) / D + 1;                            // mirror, in LDS rows
    static constexpr uint32_t ROWP = (ROW / D) * Dp;                                            // padded elements per row of ROW samples
    static constexpr uint32_t RAW_ELEMS = ((RINGD + MIRD) * Dp + 1) & ~1u;
    static constexpr uint32_t DR = 3 * GS;
    // overlapping windows: the shared FIR keeps a full value AND a truncated snapshot per output (dec + trc); windows side by side
    // (S == W): every output belongs to one window and keeps the one value that window reads (dec only)
    static constexpr bool kOverlap = S < W;
    static constexpr bool fir_ok = kOverlap ? (GeoT::kShared && GeoT::kUnrolledShared)
                                            : (S == W && GeoT::kPad == 2 && T % 4 == 0 && GeoT::b0 % 2 == 0 && D % 4 == 0 && T / 4 > 3 && (T / 2) % 4 == 0);
    static constexpr bool ok = fir_ok && GS <= 256 && GS >= 1 && N % ROW == 0 && ROW % D == 0 && D % SPL == 0 &&
                               W <= 64 * 16 && f0 >= 1 && f0 <= GS && f0 > W - S && ntrunc <= S && MIRD * D <= ROW && (G - 1) * S + W <= 2 * GS;
    static constexpr bool kWrite = (GeoT::kFlags & kGeoWriteSink) != 0;          // no FFT stage, no output ring
    static_assert(!kWrite || !kOverlap, "the write sink's sub-blocks lie side by side");
    // the two transform buffers (base pass of step s beside the layers of step s - 1) start on a 256-byte boundary: their swizzled layout
    // (SparkSwz) XORs into LDS byte addresses
    static constexpr uint32_t FBX_OFF = (RAW_ELEMS + (kOverlap ? 2u : 1u) * DR + 31u) & ~31u;
    static constexpr uint32_t kLdsBytes = kWrite ? RAW_ELEMS * 8 + ((T + 3) & ~3u) * 4
                                                 : (FBX_OFF + 2 * G * W + W) * 8 + ((T + 3) & ~3u) * 4;
    // swizzled transform buffers: the stage's LDS traffic unswizzled is 3.7x its conflict-free cycle count (the base pass's 16-byte pieces
    // 8-way, the gather and the first layer 2- to 4-way) — a tenth of the LDS array's cycles per step in a kernel whose FIR keeps the array
    // ~80 % busy (rocprofv3, cfg5).  Needs every wave's half of a buffer on a 256-byte boundary.
    static constexpr uint32_t GH = (G + 1) / 2;
    static constexpr bool kSwzFft = !kWrite && W >= 8 && (GH * W) % 32 == 0 && (G * W) % 32 == 0;
    static constexpr uint32_t kConsumerThreads = kWrite ? 256u : 512u;
};

template <int FMT, int NCO, class GeoT, int RN_, int LB, int PT_ = kPipe3Prod>
__global__ __launch_bounds__(PT_ + ((GeoT::kFlags & kGeoWriteSink) ? 256 : 512), LB) void k_chain_pipe3s(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    using K = Pipe3S<FMT, GeoT, PT_>;
    constexpr int SPL = FT::SPL;
    constexpr bool HAS_SHIFT = NCO != 0;
    static_assert(K::ok && (uint32_t)RN_ == K::RN, "streaming three-stage kernel: geometry");
    constexpr bool kWrite = K::kWrite;
    constexpr uint32_t PT = PT_, NTHR = PT_ + K::kConsumerThreads, PW = PT_ / 64;                    // producer waves [0, PW), FIR waves [PW, PW + 4), FFT waves [PW + 4, PW + 8)
    constexpr uint32_t W = K::W, S = K::S, D = K::D, T = K::T, G = K::G, Dp = K::Dp, logW = GeoT::logW;
    constexpr uint32_t ROW = K::ROW, ROWB = ROW * FT::BPS, VECB = SPL * FT::BPS, RN = K::RN, RR = K::RR, GS = K::GS, DR = K::DR;
    const GeoT geo(P);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *raw = reinterpret_cast<float2 *>(smem);
    float2 *dec = raw + K::RAW_ELEMS, *trc = dec + DR;                             // trc exists for overlapping windows only
    float2 *fbx = raw + K::FBX_OFF;                                                // two buffers of G W points, used alternately
    float2 *twl = fbx + 2 * (size_t)G * W;
    float *tapl = reinterpret_cast<float *>(kWrite ? raw + K::RAW_ELEMS : twl + W);  // write sink: sample ring | taps, nothing else
    if (P.lds_dyn < K::kLdsBytes) return;                                              // host / kernel layout disagreement: leave the output untouched (the parity tests see it)

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        if constexpr (!kWrite) {
            const uint32_t n_tw = W - geo.base_len;
            for (uint32_t i = tid; i < n_tw; i += NTHR) twl[i] = P.tw[i];
        }
        for (uint32_t i = tid; i < T; i += NTHR) tapl[i] = P.taps[i];
    }
    __syncthreads();

    // this workgroup's run: an equal, contiguous share of the launch's tiles
    const uint64_t n_tiles = (P.n_windows + G - 1) / G;
    const uint64_t base_cnt = n_tiles / gridDim.x, rem = n_tiles % gridDim.x;
    const uint64_t t_lo = (uint64_t)blockIdx.x * base_cnt + (blockIdx.x < rem ? blockIdx.x : rem);
    const uint32_t n_steps = (uint32_t)(base_cnt + (blockIdx.x < rem ? 1u : 0u));
    if (n_steps == 0) return;                                                      // uniform over the workgroup
    const uint32_t n_iter = n_steps + (kWrite ? 2u : 4u);
    auto g_cnt_of = [&](uint64_t t) -> uint32_t {
        const uint64_t w0 = t * G, left = P.n_windows - w0;
        return left < G ? (uint32_t)left : G;
    };

    if (wave < PW) {
        // ================= producers: step s = it - 1 parks rows [(s + 1) RN, (s + 2) RN) of the run in the ring
        LaneRot lr[SPL];
        if constexpr (HAS_SHIFT) {
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const uint32_t j = tid * SPL + u;
                const double2 cs = P.jtab[j];
                lr[u].jf = (double)j; lr[u].c = cs.x; lr[u].s = cs.y;
            }
        }
        const uint32_t lane_pad = pad_index(geo, tid * SPL);
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
        const uint64_t n0 = (P.first_window + t_lo * G) * ((uint64_t)S * D);        // the run's first raw sample (a row boundary)
        auto rsrc_of = [&](uint32_t step1) {                                       // step1 = s + 1: rows [step1 RN, (step1 + 1) RN)
            const uint64_t ns = n0 + (uint64_t)step1 * K::N;
            const uint64_t end = P.src_first + P.src_count;
            const uint64_t left = ns < end ? (end - ns) * FT::BPS : 0;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * FT::BPS, 0,
                                                     left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
        };
        Vec pf[RN];
        auto load_row = [&](const decltype(rsrc_of(0)) &rsrc, int i) {
            constexpr int aux = ct_load_aux(GeoT::kFlags);                          // every row is read exactly once: all of them may go non-temporal
            if constexpr (sizeof(Vec) == 16) {
                const v4u_t w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(tid * VECB), (int)(i * ROWB), aux);
                pf[i].x = w.x; pf[i].y = w.y; pf[i].z = w.z; pf[i].w = w.w;
            } else {
                const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(tid * VECB), (int)(i * ROWB), aux);
                pf[i].x = w.x; pf[i].y = w.y;
            }
        };
        double rt_touch = 0.0;
        TileGeo gl{};
        gl.tile_raw = ROW;
        __builtin_amdgcn_s_setprio(0);
        {
            const auto rsrc = rsrc_of(0);
#pragma unroll
            for (int i = 0; i < (int)RN; ++i) load_row(rsrc, i);
        }
        uint32_t slot0 = 0;                                                         // ring row of the step's first row
        for (uint32_t it = 0; it < n_iter; ++it) {
            if (it <= n_steps && !QD_DBG(P, 256)) {
                const uint32_t step1 = it;                                          // s + 1
                const uint32_t pf1 = it < n_steps ? it + 1 : it;                    // last step: harmless re-loads
                const auto rsrc = rsrc_of(pf1);
                const uint64_t ns = n0 + (uint64_t)step1 * K::N;
                const_f64_p rows = (const_f64_p)(uintptr_t)(P.rowtab + (ns / ROW - P.rowtab_row0));
                if constexpr (HAS_SHIFT) {                                          // L2 touch of the next step's row bases
                    uint32_t r = tid < RN ? tid : RN - 1;
                    asm volatile("" : "+v"(r));
                    rt_touch += P.rowtab[(n0 + (uint64_t)pf1 * K::N) / ROW - P.rowtab_row0 + r].c;
                }
                RowBase rb_next{};
                if constexpr (HAS_SHIFT) rb_next = load_rowbase_at(rows, 0);
#pragma unroll
                for (int i = 0; i < (int)RN; ++i) {
                    const Vec v = pf[i];
                    const RowBase rb = rb_next;
                    if constexpr (HAS_SHIFT) { if (i + 1 < (int)RN) rb_next = load_rowbase_at(rows, i + 1); }
                    uint32_t slot = slot0 + (uint32_t)i;
                    if (slot >= RR) slot -= RR;
                    float2 *dst = raw + (size_t)slot * K::ROWP;
                    float2 *mir = slot == 0 ? raw + (size_t)K::RINGD * Dp : nullptr;    // the ring's first samples once more behind its end
                    process_row<FMT, PT, NCO, true>(P, geo, gl, 0, tid, v, rb, lr, lane_pad, nullptr, dst, mir, K::MIRD * D);
                    __builtin_amdgcn_sched_barrier(0);
                    load_row(rsrc, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot0 += RN;
                if (slot0 >= RR) slot0 -= RR;
            }
            __syncthreads();
        }
        if (P.dbg == 0xdeadbeefu) reinterpret_cast<double *>(P.out)[tid] = rt_touch;   // never true: keeps rt_touch live
    } else if (wave < PW + 4) {
        // ================= the shared FIR: step s = it - 2 computes the G S outputs its rows completed (the cold start: f0)
        __builtin_amdgcn_s_setprio(2);
        constexpr uint32_t ntrunc = K::ntrunc;
        for (uint32_t it = 0; it < n_iter; ++it) {
            if (it >= 1 && it <= n_steps + 1) {
                const uint32_t q_lo = it == 1 ? 0u : K::f0 + (it - 2) * GS, cnt = it == 1 ? K::f0 : GS;
                uint32_t l = tid - PT;
                asm volatile("" : "+v"(l));
                if (l < cnt && !QD_DBG(P, 64)) {
                    const uint32_t q = q_lo + l;                                    // run-local output index
                    uint32_t jmax = T;
                    if constexpr (kWrite) {
                        // truncation relative to the read_at block of blk_len outputs the sub-block lies in (src/filter.rs:68-83 over
                        // do_write's blocks); every other output is a full chain, whatever sub-block it ends
                        const uint64_t sub = P.first_window + t_lo * G + (q >> logW);
                        const uint32_t kb = ((uint32_t)sub & P.blk_sub_mask) * W + (q & (W - 1));
                        const uint32_t jm = (P.blk_len - kb) * D + T / 2;
                        if (jm < T) jmax = jm;
                    } else
                    if constexpr (K::kOverlap) {
                        if (q + ntrunc >= W) {                                      // may be in the truncated tail of a window of the run
                            const uint32_t g = (q - (W - ntrunc)) / S, k = q - g * S;
                            if (k < W) { const uint32_t jm = (W - k) * D + T / 2; if (jm < T) jmax = jm; }
                        }
                    } else {
                        const uint32_t k = q & (W - 1);                             // S == W: position in the one window the output belongs to
                        const uint32_t jm = (W - k) * D + T / 2;
                        if (jm < T) jmax = jm;
                    }
                    const uint32_t drow = (q + geo.a0) % K::RINGD;
                    const float2 *rowp = raw + (size_t)drow * Dp;
                    float2 snap = make_float2(0.f, 0.f);
                    // The taps sit behind the sample ring, ~150 KB into LDS — past the 16-bit offset field of a DS instruction.  Left as a
                    // compile-time address, every one of the T/4 broadcast tap reads gets an address register of its own, materialised
                    // in an SGPR outside the loop (a hundred of them, most spilled to lanes of a VGPR) and moved to a VGPR per read:
                    // ~100 VALU + ~100 SALU instructions per pass on the wave whose pass IS the step's critical path.  One opaque base
                    // register instead: every tap read is base + immediate.
                    typedef const float __attribute__((address_space(3))) *lds_f32_p;
                    uint32_t taps_off = (uint32_t)(uintptr_t)(lds_f32_p)tapl;            // the LDS byte offset, in a VGPR the compiler cannot see through
                    asm volatile("" : "+v"(taps_off));
                    const float *taps_v = (const float *)(lds_f32_p)(uintptr_t)taps_off;
                    const float2 full = fir_pair<GeoT, true>(rowp, jmax, taps_v, &snap);
                    const uint32_t pos = q % DR;
                    if constexpr (kWrite) {
                        // do_write / LowPass::read_at output (src/lib.rs:206-209): the decimated cf32 samples themselves, in stream order
                        const uint64_t qa = (t_lo * G << logW) + q;                // output index within the launch
                        if (qa < (P.n_windows << logW))
                            reinterpret_cast<float2 *>(P.out)[((P.first_window - P.out_window0) << logW) + qa] = jmax < T ? snap : full;
                    } else
                    if constexpr (K::kOverlap) {
                        dec[pos] = full;
                        if (jmax < T) trc[pos] = snap;
                    } else {
                        dec[pos] = jmax < T ? snap : full;
                    }
                }
            }
            __syncthreads();
        }
    } else if constexpr (!kWrite) {
        // ================= the transform, two waves per role, each role on half of the tile's windows:
        //   waves 0, 1: gather + the BASE butterflies of step s = it - 3 into transform buffer s & 1;
        //   waves 2, 3: the radix-4 layers, |X| and the epilogue of step s = it - 4 out of buffer s & 1.
        // Round 3 gave every wave G / 4 windows from gather to store: the sixteen-point base butterfly of a 64-point window is ONE
        // task per quarter window, so each of the four waves walked its ~250 instructions with 16 of 64 lanes active (and the last
        // wave with 8) — 1000 wave-instructions per step for work that fills 56 lanes once.  Split by ROLE a wave has twice the
        // windows in the pass it runs (28 base tasks; 112 butterflies and 448 bins: two and seven full rounds) and the transform
        // stage issues ~45 % fewer instructions per step; the price is one more step of latency per run and a second G W buffer.
        __builtin_amdgcn_s_setprio(1);
        const uint32_t v = wave - (PW + 4);
        constexpr uint32_t GH = (G + 1) / 2;
        const uint32_t g0 = (v & 1u) * GH;
        uint32_t rot = 0;                                                          // (step mod 3) G S: where the step's outputs start in the output ring
        for (uint32_t it = 0; it < n_iter; ++it) {
            if (v < 2) {
                if (it >= 3 && it - 3 < n_steps) {
                    const uint32_t s = it - 3;
                    const uint32_t g_cnt = g_cnt_of(t_lo + s);
                    const uint32_t g1 = (g0 + GH < g_cnt) ? g0 + GH : g_cnt;
                    if (g0 < g1 && !QD_DBG(P, 128)) {
                        float2 *fbw = fbx + (size_t)(s & 1u) * G * W + (size_t)g0 * W;
                        uint32_t lane = tid & 63u;
                        asm volatile("" : "+v"(lane));
                        constexpr uint32_t log_width = 2 * GeoT::layers;
                        const uint32_t n_o = (g1 - g0) << logW;
                        if constexpr (K::kSwzFft) {
                            using SZ = SparkSwzFn<SparkSwz<W, 1>>;
                            const uint32_t fb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)fbw;
                            for (uint32_t o = lane; o < n_o; o += 64) {
                                const uint32_t gl_ = o >> logW, k = o & (W - 1);
                                uint32_t pos = rot + (g0 + gl_) * S + k;
                                pos = pos >= DR ? pos - DR : pos;
                                const bool tr = K::kOverlap && (W - k) * D + T / 2 < T;
                                const float2 val = dec[pos + (tr ? DR : 0u)];
                                const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                                const uint32_t idx = (gl_ << logW) + yy + (rev4(xx, GeoT::layers) << GeoT::log_base);
                                spark_st2(fb + ((idx ^ SZ::delta(idx)) << 3), val);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            spark_fft_swz<GeoT, K::GH * W, 1, 1, true>(P, twl, fb, lane, g1 - g0);
                        } else {
                        for (uint32_t o = lane; o < n_o; o += 64) {
                            const uint32_t gl_ = o >> logW, k = o & (W - 1);
                            // ring position (s G S + g S + k) mod 3 G S: g S + k < 2 G S (Pipe3S::ok), the step's base rotates through 0, GS, 2GS
                            uint32_t pos = rot + (g0 + gl_) * S + k;
                            pos = pos >= DR ? pos - DR : pos;
                            const bool tr = K::kOverlap && (W - k) * D + T / 2 < T;       // the window's truncated tail reads the snapshot ring (dec + DR)
                            const float2 val = dec[pos + (tr ? DR : 0u)];
                            const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                            fbw[(gl_ << logW) + yy + (rev4(xx, geo.layers) << geo.log_base)] = val;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        wave_fft_epilogue_fn<GeoT, 0, 1>(P, geo, twl, fbw, 0, g1 - g0, tid);
                        }
                    }
                }
                if (it >= 3) { rot += GS; rot = rot >= DR ? rot - DR : rot; }
            } else if (it >= 4) {
                const uint32_t s = it - 4;
                const uint64_t t = t_lo + s;
                const uint32_t g_cnt = g_cnt_of(t);
                const uint32_t g1 = (g0 + GH < g_cnt) ? g0 + GH : g_cnt;
                if (g0 < g1 && !QD_DBG(P, 128)) {
                    float2 *fbp = fbx + (size_t)(s & 1u) * G * W + (size_t)g0 * W;
                    if constexpr (K::kSwzFft) {
                        using SZ = SparkSwzFn<SparkSwz<W, 1>>;
                        const uint32_t fb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)fbp;
                        uint32_t lane = tid & 63u;
                        asm volatile("" : "+v"(lane));
                        spark_fft_swz<GeoT, K::GH * W, 1, 2, true>(P, twl, fb, lane, g1 - g0);
                        // the epilogue of wave_fft_epilogue_fn over the swizzled buffer (src/fft.rs:48-61, :86-97)
                        const uint64_t wrel = P.first_window + t * G + g0 - P.out_window0;
                        const uint32_t n_out_s = (g1 - g0) << logW;
                        if (P.epi == 2) {
                            constexpr uint32_t KB = (K::GH * W + 63) / 64;
                            float nm[KB];
#pragma unroll
                            for (uint32_t k = 0; k < KB; ++k) { const uint32_t o = lane + 64 * k; nm[k] = o < n_out_s ? norm_ref(spark_ld2(fb + ((o ^ SZ::delta(o)) << 3))) : 0.f; }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                            float *nb = reinterpret_cast<float *>(fbp);
#pragma unroll
                            for (uint32_t k = 0; k < KB; ++k) { const uint32_t o = lane + 64 * k; if (o < n_out_s) nb[o] = nm[k]; }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                            for (uint32_t wl = lane; wl < g1 - g0; wl += 64) {
                                const float *q = nb + (wl << logW);
                                float first = 0.f, second = 0.f;
                                for (uint32_t k = 0; k < W / 2; ++k) first = first + q[k];
                                for (uint32_t k = W / 2; k < W; ++k) second = second + q[k];
                                reinterpret_cast<uint8_t *>(P.out)[wrel + wl] = first < second ? 0 : 1;
                            }
                        } else {
                            float *outf = reinterpret_cast<float *>(P.out) + (wrel << logW);
                            uint8_t *outb = reinterpret_cast<uint8_t *>(P.out) + (wrel << logW);
                            for (uint32_t o = lane; o < n_out_s; o += 64) {
                                const uint32_t e = o ^ (W >> 1);
                                const float nm = norm_ref(spark_ld2(fb + ((e ^ SZ::delta(e)) << 3)));
                                if (P.epi == 0) outf[o] = nm;
                                else outb[o] = glyph_of<GeoT>(P, nm);
                            }
                        }
                    } else
                    wave_fft_epilogue_fn<GeoT, (GH * W + 63) / 64, 2>(P, geo, twl, fbp, P.first_window + t * G + g0, g1 - g0, tid);
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- chains WITHOUT a lowpass: the wave-local kernel (FLAGS bit 19)
//
// `from F [shift] sparkfft` (README example 1, BASELINE configs[0]; src/fft.rs:28-65 straight over Shift / SampleFile): every input
// sample is an FFT input, windows lie side by side (stride == width), so a window is a pure function of W consecutive samples and
// nothing is shared between windows at all.  k_chain serves that shape with its tile machinery — phase 1 into a raw tile, a barrier,
// a gather into transform order, a barrier, the base butterflies, a barrier per Radix4 layer, the epilogue — and runs at the SUM of
// its HBM time and its arithmetic time (16 GiB cf32, W = 128: 6.7 ms where either alone is ~3; profiles/r03/nofir_rate.log).
// Here a WAVE owns a tile of TS = NCH * 64 * SPL consecutive samples (1024 for every format: G = 1024 / W windows) from load to
// store, and no workgroup barrier exists after the prologue:
//   * one coalesced row load per chunk (64 lanes x 16 B, or x 8 B for the 8-bit formats), the NEXT tile's chunks prefetched into
//     the registers the current ones vacate (non-temporal: every byte is read once), through a per-tile buffer descriptor whose
//     range check covers the slab end and a short last tile;
//   * unpack and NCO multiply as in k_chain (same process arithmetic: nco_mul_n, cmul_pk, the reciprocal unpack), the shifted
//     sample written STRAIGHT to its place in rustfft's transposed input order (bitreversed_transpose) in the wave's own LDS slice —
//     there is no raw tile and no gather pass;
//   * the Radix4 passes, |X| and the epilogue wave-local (wave_fft_epilogue_fn: LDS operations of one wave execute in order, the
//     passes are separated by compiler-level fences); with 1024 samples per tile the base butterflies fill every lane for W >= 16.
// Sixteen independent waves per CU, each with its next tile (8 KiB cf32) in flight, hide the HBM latency that k_chain's four
// workgroups hid with a tile of prefetch behind a chain of barriers.  Products, order and roundings are k_chain's, so the output
// is bit-identical to the generic kernel's (tests/test_gpu_robustness.py::test_wave_local_kernel_equals_generic) and to the oracle.
// NCO row geometry: rows of 512 samples for every format (ChainParams::rowtab, jtab with 512 entries); a tile is two rows, a chunk
// the quarter (cf32) or half (8-bit, cs16) of a row, so the lane constants of chunk c are those of quarter c % RQ: RQ * SPL = 8
// (cos, sin) pairs per lane, kept in registers.
constexpr uint32_t kGeoSpark = 524288;       // FLAGS bit 19 (reported in qd_plan_info.kernel_flags)
constexpr uint32_t kSparkRow = 512;          // samples per NCO row of this kernel, every format
constexpr uint32_t kSparkMaxW = 1024;

// Which tiles a wave takes.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with an L2 of its own whose
// channels interleave the address space in 4 KiB steps.  A chip-wide grid-stride walk hands XCD x the 32 KiB chunks x, x + 8, x + 16 ...
// of the stream — a quarter-MiB stride under which an XCD only ever touches HALF of its L2 channels (16 GiB cf32, W = 128: 3.0 TB/s
// of reads with every arithmetic phase ablated, profiles/r04/spark_ablate.log).  So each XCD walks one contiguous eighth of the tile
// range and its waves sit on neighbouring tiles: a few MiB of contiguous addresses in flight per XCD, every channel busy.
struct SparkWalk {
    uint64_t first, stride, end;
    __device__ __forceinline__ SparkWalk(uint64_t n_tiles, uint32_t wave) {
        constexpr uint32_t WPG = kThreads / 64;
        if ((gridDim.x & 7u) == 0 && n_tiles >= 64) {
            const uint64_t n8 = (n_tiles + 7) / 8, lo = (uint64_t)(blockIdx.x & 7u) * n8;
            first = lo + (uint64_t)(blockIdx.x >> 3) * WPG + wave;
            stride = (uint64_t)(gridDim.x >> 3) * WPG;
            end = lo + n8 < n_tiles ? lo + n8 : n_tiles;
            if (lo >= n_tiles) { first = n_tiles; end = n_tiles; }
        } else {
            first = (uint64_t)blockIdx.x * WPG + wave; stride = (uint64_t)gridDim.x * WPG; end = n_tiles;
        }
    }
};

template <class GeoT, bool F = GeoT::kFixed> struct W_CT_GE8 { static constexpr bool value = false; };      // a compile-time width of at least 8
template <class GeoT> struct W_CT_GE8<GeoT, true> { static constexpr bool value = GeoT::W >= 8; };
template <int FMT> struct SparkTraits {
    using FT = FmtTraits<FMT>;
    static constexpr uint32_t SPL = FT::SPL, CH = 64u * SPL, RQ = kSparkRow / CH;       // chunk: one wave-wide load; RQ chunks per NCO row
};

template <int FMT, int NCO, class GeoT, int NCH, int LB, int EPI = -1 /* plan-time builds: the plan's sink, compile-time (norms 0 / glyph 1: the lean epilogue below); -1: P.epi at run time */>
__global__ __launch_bounds__(kThreads, LB) void k_spark(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    using Vec = typename FT::Vec;
    using ST = SparkTraits<FMT>;
    constexpr int SPL = FT::SPL;
    constexpr bool HAS_SHIFT = NCO != 0;
    constexpr uint32_t CH = ST::CH, RQ = ST::RQ, TS = (uint32_t)NCH * CH, CHB = CH * FT::BPS, VECB = SPL * FT::BPS;
    static_assert(TS % kSparkRow == 0 && (NCH % RQ) == 0, "a tile is a whole number of NCO rows");
    const GeoT geo(P);
    const uint32_t W = geo.W, logW = geo.logW;
    const uint32_t G = TS >> logW;                                         // windows per tile (host: P.G == G, W <= TS)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *twl = reinterpret_cast<float2 *>(smem);                        // radix-4 layer twiddles (< W entries), shared by the four waves
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Plan-time builds with a shift read the lane constants out of an LDS copy of the lane table (one ds_read_b128 per sample) instead of
    // holding RQ * SPL (cos, sin) pairs — 32 registers — across the tile loop: that is what kept them at three waves per SIMD.
    constexpr bool kLdsLane = HAS_SHIFT && GeoT::kFixed;
    double2 *jt = reinterpret_cast<double2 *>(twl + (W < 32u ? 32u : W));      // (twiddle area of at least 256 bytes: the transform buffers start on 256-byte boundaries)
    float2 *fbw = twl + (W < 32u ? 32u : W) + (kLdsLane ? 2u * kSparkRow : 0u) + (size_t)wave * TS;           // this wave's transform buffer: TS complex samples
    {
        const uint32_t n_tw = W - geo.base_len;
        for (uint32_t i = tid; i < n_tw; i += kThreads) twl[i] = P.tw[i];
        if constexpr (kLdsLane) { for (uint32_t i = tid; i < kSparkRow; i += kThreads) jt[i] = P.jtab[i]; }
    }
    __syncthreads();                                                       // the only workgroup barrier

    // per-lane constants: where each of the lane's NCH * SPL samples of a tile goes (bitreversed_transpose::<4>(base_len, ..):
    // out[y + rev(x) * base] = in[x + y * width]), and the NCO lane rotations of the RQ row quarters
    uint32_t pos[NCH * SPL];
    {
        const uint32_t log_width = 2 * geo.layers;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const uint32_t m = (uint32_t)c * CH + lane * SPL + (uint32_t)u, k = m & (W - 1);
                const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
                pos[c * SPL + u] = (m & ~(W - 1)) + yy + (rev4(xx, geo.layers) << geo.log_base);
            }
    }
    // plan-time builds with the sink known (the lean path below), W >= 8: the transform buffer in the swizzled layout (SparkSwz)
    constexpr bool kLeanEpi = GeoT::kFixed && (EPI == 0 || EPI == 1 || EPI == 2);
    const uint32_t fb_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)fbw;      // LDS byte offset of the wave's buffer
    if (kLeanEpi && (fb_off & 255u)) __builtin_trap();                      // (dynamic LDS starts at offset 0: never taken; the swizzled addressing relies on it)
    if constexpr (kLeanEpi) {
        if constexpr (GeoT::W >= 8) {
            using SZ = SparkSwzFn<SparkSwz<GeoT::W, (uint32_t)SPL>>;
#pragma unroll
            for (int i = 0; i < NCH * SPL; ++i) pos[i] ^= SZ::delta(pos[i]);
        }
    }
    // (cos, sin)(j * ratio) of the lane's sample slots in each row quarter; the slot index itself is kept for quarter 0 only — the
    // quarter's offset q * CH goes into the row's sample count instead (integers below 2^53: the same double whichever way they add up)
    double2 lcs[HAS_SHIFT && !kLdsLane ? RQ * SPL : 1];
    double ljf[HAS_SHIFT ? SPL : 1];
    if constexpr (HAS_SHIFT) {
        if constexpr (!kLdsLane) {
#pragma unroll
            for (uint32_t q = 0; q < RQ; ++q)
#pragma unroll
                for (int u = 0; u < SPL; ++u) lcs[q * SPL + u] = P.jtab[q * CH + lane * SPL + (uint32_t)u];
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) ljf[u] = (double)(lane * SPL + (uint32_t)u);
    }

    const uint64_t n_tiles = (P.n_windows + G - 1) / G;
    SparkWalk walk(n_tiles, wave);
    uint64_t tile = walk.first;
    const uint64_t n_waves = walk.stride, tile_end = walk.end;
    if (tile >= tile_end) return;                                          // (after the barrier; wave-uniform)
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto rsrc_of = [&](uint64_t t) {
        const uint64_t ns = (P.first_window + t * G) << logW, end = P.src_first + P.src_count;
        const uint64_t left = ns < end ? (end - ns) * FT::BPS : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * FT::BPS, 0,
                                                 left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
    };
    Vec pf[NCH];
    auto load_chunk = [&](const decltype(rsrc_of(0)) &rsrc, int c) {
        constexpr int aux = 2;                                             // nt: the stream is read once
        if constexpr (sizeof(Vec) == 16) {
            const v4u_t w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane * VECB), (int)(c * CHB), aux);
            pf[c].x = w.x; pf[c].y = w.y; pf[c].z = w.z; pf[c].w = w.w;
        } else {
            const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(lane * VECB), (int)(c * CHB), aux);
            pf[c].x = w.x; pf[c].y = w.y;
        }
    };
    {
        const auto rsrc = rsrc_of(tile);
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_chunk(rsrc, c);
    }
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // The lean epilogue (plan-time builds with the sink known: see k_spark2 for the reasoning): no sink dispatch and no store branch in
    // the tile loop — the output goes through a per-tile buffer descriptor that ends with the tile's last valid window —, four |X| per
    // IEEE-path test, non-temporal stores, and the loop's entry edge issues as many (dropped) stores as a tile does so that the waits
    // for the prefetched chunks stay counted instead of vmcnt(0).
    // (the bucket sink, EPI 2, takes the lean path from W = 8 on: swizzled transform, norms parked as floats, one lane per window sums the
    // halves in order, the digit through the same range-checked descriptor — one store instruction per 64 windows of the tile)
    constexpr bool kLean = GeoT::kFixed && (EPI == 0 || EPI == 1 || (EPI == 2 && GeoT::kFixed && W_CT_GE8<GeoT>::value));
    constexpr uint32_t NB = TS / 64;                                       // bins per lane and tile
    if constexpr (kLean) {
        const auto none = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out), 0, 0, 0x00020000);
        constexpr uint32_t NDUM = EPI == 2 ? ((TS >> GeoT::logW) + 63) / 64 : NB;
#pragma unroll
        for (uint32_t q = 0; q < NDUM; ++q) {
            if constexpr (EPI == 0) __builtin_amdgcn_raw_buffer_store_b32(0u, none, (int)(lane * 4 + q * 256), 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, none, (int)(lane + q * 64), 0, 2);
        }
    }
    while (true) {
        const uint64_t tile_n = tile + n_waves;
        const auto rsrc_n = rsrc_of(tile_n < tile_end ? tile_n : tile);     // last tile of this wave: harmless re-loads
        const uint64_t w0 = P.first_window + tile * G, left_w = P.first_window + P.n_windows - w0;
        const uint32_t g_cnt = left_w < G ? (uint32_t)left_w : G;
        const_f64_p rows = nullptr;
        if constexpr (HAS_SHIFT) rows = (const_f64_p)(uintptr_t)(P.rowtab + (((w0 << logW) / kSparkRow) - P.rowtab_row0));
        RowBase rb{};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const Vec v = pf[c];
            if constexpr (HAS_SHIFT) { if (c % (int)RQ == 0) rb = load_rowbase_at(rows, c / (int)RQ); }
            float2 x[SPL];
            if constexpr (FMT == 0) {
                x[0] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                x[1] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
            } else if constexpr (FMT == 1) {
                const uint32_t a = v.x ^ 0x80808080u, b = v.y ^ 0x80808080u;
                x[0] = make_float2(unpack_cs8_at(a, 0), unpack_cs8_at(a, 1));
                x[1] = make_float2(unpack_cs8_at(a, 2), unpack_cs8_at(a, 3));
                x[2] = make_float2(unpack_cs8_at(b, 0), unpack_cs8_at(b, 1));
                x[3] = make_float2(unpack_cs8_at(b, 2), unpack_cs8_at(b, 3));
            } else if constexpr (FMT == 2) {
                x[0] = make_float2(unpack_cu8_at(v.x, 0), unpack_cu8_at(v.x, 1));
                x[1] = make_float2(unpack_cu8_at(v.x, 2), unpack_cu8_at(v.x, 3));
                x[2] = make_float2(unpack_cu8_at(v.y, 0), unpack_cu8_at(v.y, 1));
                x[3] = make_float2(unpack_cu8_at(v.y, 2), unpack_cu8_at(v.y, 3));
            } else {
                x[0] = make_float2(unpack_cs16(v.x & 0xffffu), unpack_cs16(v.x >> 16));
                x[1] = make_float2(unpack_cs16(v.y & 0xffffu), unpack_cs16(v.y >> 16));
                x[2] = make_float2(unpack_cs16(v.z & 0xffffu), unpack_cs16(v.z >> 16));
                x[3] = make_float2(unpack_cs16(v.w & 0xffffu), unpack_cs16(v.w >> 16));
            }
            if constexpr (HAS_SHIFT) {
                float2 m[SPL];
                const int q = c % (int)RQ;
                LaneRot lr[SPL];
#pragma unroll
                for (int u = 0; u < SPL; ++u) {
                    const double2 cs = kLdsLane ? jt[(uint32_t)q * CH + lane * SPL + (uint32_t)u] : lcs[kLdsLane ? 0 : q * SPL + u];
                    lr[u].jf = ljf[u]; lr[u].c = cs.x; lr[u].s = cs.y;
                }
                RowBase rbq = rb;
                rbq.nf = rb.nf + (double)(q * (int)CH);
                nco_mul_n<NCO == 2, SPL>(rbq, lr, P.ratio, m);
#pragma unroll
                for (int u = 0; u < SPL; ++u) x[u] = cmul_pk(x[u], m[u]);       // buf[i] *= mul (src/shift.rs:51)
            }
#pragma unroll
            for (int u = 0; u < SPL; ++u) fbw[pos[c * SPL + u]] = x[u];
            __builtin_amdgcn_sched_barrier(0);                              // refill slot c only after chunk c is consumed (see k_chain)
            load_chunk(rsrc_n, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        wsync();
        if constexpr (kLean) {
            // every window of the tile (rows past the slab read as zeros): compile-time trip counts
            if constexpr (GeoT::W >= 8) spark_fft_swz<GeoT, TS, (uint32_t)SPL>(P, twl, fb_off, lane);
            else wave_fft_epilogue_fn<GeoT, 0, 3>(P, geo, twl, fbw, w0, G, tid);
            if constexpr (EPI == 2) {
                // freq_levels (src/fft.rs:86-97): |X| of every bin in its own order, parked as floats over the transform buffer, then one lane
                // per window sums the two halves sequentially
                using SZB = SparkSwzFn<SparkSwz<GeoT::W, (uint32_t)SPL>>;
                uint32_t lo = lane;
                asm volatile("" : "+v"(lo));
                const uint32_t b0 = fb_off + ((lo ^ SZB::delta(lo)) << 3);
                float nmv[NB];
#pragma unroll
                for (uint32_t k = 0; k < NB; ++k) {
                    const uint32_t ec = 64 * k;
                    nmv[k] = norm_ref(spark_ld2((b0 ^ (SZB::delta(ec) << 3)) + (ec << 3)));
                }
                wsync();
                float *nb = reinterpret_cast<float *>(fbw);
#pragma unroll
                for (uint32_t k = 0; k < NB; ++k) nb[lo + 64 * k] = nmv[k];
                wsync();
                constexpr uint32_t GT = TS >> GeoT::logW;                   // windows per tile
                const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out) + (w0 - P.out_window0), 0, g_cnt, 0x00020000);
#pragma unroll
                for (uint32_t wl0 = 0; wl0 < GT; wl0 += 64) {
                    const uint32_t wl = wl0 + lo;
                    uint8_t digit = 0;
                    if (wl < GT) {
                        const float *q = nb + (wl << GeoT::logW);
                        float first = 0.f, second = 0.f;
                        for (uint32_t k = 0; k < GeoT::W / 2; ++k) first = first + q[k];
                        for (uint32_t k = GeoT::W / 2; k < GeoT::W; ++k) second = second + q[k];
                        digit = first < second ? 0 : 1;
                    }
                    __builtin_amdgcn_raw_buffer_store_b8(digit, orsrc, (int)(wl < GT ? wl : 0xffffffu), 0, 2);      // lanes without a window: past the descriptor's end
                }
            } else {
            constexpr uint32_t OBW = EPI == 0 ? 4u * GeoT::W : GeoT::W;
            const uint32_t RS = P.out_row_stride ? P.out_row_stride : 1u;      // rows between this launch's windows (interleaved launches of overlapping windows)
            const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out) + (w0 - P.out_window0) * RS * OBW, 0,
                                                                 g_cnt ? ((g_cnt - 1) * RS + 1) * OBW : 0u, 0x00020000);
            uint32_t lo = lane;
            asm volatile("" : "+v"(lo));
            // element (lo + 64 k) ^ (W / 2) of the swizzled buffer: the lane's bits and the compile-time ones swizzle separately (delta is linear)
            using SZE = SparkSwzFn<SparkSwz<(GeoT::W >= 8 ? GeoT::W : 0u), (uint32_t)SPL>>;
            const uint32_t el = lo ^ ((GeoT::W >> 1) & 63u), e0 = fb_off + ((el ^ SZE::delta(el)) << 3);
#pragma unroll
            for (uint32_t k0 = 0; k0 < NB; k0 += 4) {
                float nm[4];
                bool sl[4];
                float2 xv[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {                          // fftshift: bin (b + W/2) mod W of the same window
                    constexpr uint32_t hi_w = (GeoT::W >> 1) & ~63u;
                    const uint32_t ec = (64 * (k0 + q)) ^ hi_w;
                    xv[q] = spark_ld2((e0 ^ (SZE::delta(ec) << 3)) + (ec << 3));      // XOR value: byte bits 4 ... 7, below the 256-byte-aligned base; ec: an instruction offset
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) nm[q] = norm_fast(xv[q].x, xv[q].y, sl[q]);
                if (__builtin_expect(sl[0] | sl[1] | sl[2] | sl[3], 0)) {
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) if (sl[q]) nm[q] = norm_ieee(xv[q].x, xv[q].y);
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t o = lo + 64 * (k0 + q);
                    const uint32_t oo = (((o >> GeoT::logW) * RS) << GeoT::logW) + (o & (GeoT::W - 1));      // window o / W lands RS rows apart
                    if constexpr (EPI == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nm[q]), orsrc, (int)(oo * 4), 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b8(glyph_of<GeoT>(P, nm[q]), orsrc, (int)oo, 0, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        } else
        wave_fft_epilogue_fn<GeoT, TS / 64>(P, geo, twl, fbw, w0, g_cnt, tid);
        wsync();                                                            // the next tile's samples overwrite what the epilogue read
        if (tile_n >= tile_end) break;
        tile = tile_n;
    }
}


// ---------------------------------------------------------------- k_spark2: the same chain with the first FFT pass out of REGISTERS (FLAGS bits 19 + 20)
//
// k_spark parks every sample in LDS, then reads the base butterflies' inputs back: 16 points per lane written with a lane stride that
// conflicts 2-way, read with one that conflicts 4- to 8-way, for the one pass of the transform whose inputs are nothing but the loaded
// samples in another order.  rustfft's Radix4 starts from the transposed input (bitreversed_transpose: position y + rev4(x) * base
// holds input x + y * width, width = W / base = 4^layers): base butterfly `x` works on inputs x, x + width, x + 2 width, ... — a COLUMN
// of the window seen as a base x width matrix.  So here a lane loads a column pair directly: lane (g, xp) of a wave issues `base` row
// loads of 16 bytes, row y at byte (g W + y width + 2 xp) * 8 of the tile; the lanes of one window cover width * 8 contiguous bytes per
// instruction (128 B for W = 128, 512 B for W = 512 / 1024 — whole cache lines, each fetched by exactly one instruction).  The two
// base butterflies of the lane run on those registers, their outputs go to LDS once, the radix-4 layers run in place (every lane's
// butterflies unrolled: their reads are issued together), and the LAST layer never stores: a butterfly's four results are bins
// i, i + W/4, i + W/2, i + 3W/4, i.e. after the fftshift four outputs of the same window, so |X| (and the glyph) go from registers to
// HBM.  The registers the rows arrived in are refilled with the NEXT tile's rows as soon as the base pass has consumed them — that
// is the prefetch, with no registers of its own.  Layer twiddles: a lane's butterflies t = lane + 64 k share i = t mod cols while
// cols <= 64, so those twiddles live in registers; wider layers read them from LDS.
// With a shift: the lane's samples sit at 2 base different places j of their NCO row (rows of 512 samples), so the lane constants
// (cos, sin)(j ratio) come from an LDS copy of the lane table (one ds_read_b128 per sample) instead of 64 registers, the row base is
// selected per lane from the tile's rows; same table entries, same operations as k_chain: bit-identical output.
// W = 128 ... 1024 (width 16 or 64), every format (the integer formats' rows arrive packed: 4 or 8 bytes per lane and row); everything
// else stays on k_spark.  Tile: 64 lanes x 2 columns x base rows = 1024 samples (base 8: W = 128, 512) or 2048 (base 16: W = 256, 1024).
constexpr uint32_t kGeoSparkReg = 1048576;   // FLAGS bit 20

// LDS swizzle of k_spark2's transform buffer: the same GF(2) maps as SparkSwz, searched over this kernel's LDS instructions (the base
// pass's 16-byte piece writes of lane (g, xp), the layers' reads and writes; scripts/lds_swizzle_search2.py).  Base 16 (W = 256, 1024):
// the hand-made swizzle the kernel started with, conflict-free in the model and on the counters; base 8 (W = 128, 512): the piece writes
// were 4-way (lane-group cycles per tile 384 -> 192 and 480 -> 288).
template <uint32_t W> struct Spark2Swz { static constexpr uint32_t m[4] = {0, 0, 0, 0}; };
template <> struct Spark2Swz<128> { static constexpr uint32_t m[4] = {0x40, 0x10, 0x20, 0x40}; };
template <> struct Spark2Swz<256> { static constexpr uint32_t m[4] = {0x80, 0x10, 0x20, 0x40}; };
template <> struct Spark2Swz<512> { static constexpr uint32_t m[4] = {0x100, 0x20, 0x60, 0x40}; };
template <> struct Spark2Swz<1024> { static constexpr uint32_t m[4] = {0x200, 0x40, 0x80, 0x40}; };
template <int FMT> struct Spark2Raw { using type = uint32_t; };          // a row's column pair as loaded: one dword (cs8 / cu8) ...
template <> struct Spark2Raw<0> { typedef unsigned type __attribute__((ext_vector_type(4))); };       // ... four (cf32) ...
template <> struct Spark2Raw<3> { typedef unsigned type __attribute__((ext_vector_type(2))); };       // ... two (cs16)
template <class GeoT> struct Spark2 {
    static constexpr uint32_t W = GeoT::W, base = GeoT::base_len, layers = GeoT::layers, width = W / base;
    static constexpr uint32_t LPW = width / 2, GW = 64 / (LPW ? LPW : 1), TS = GW * W, NBF = TS / 256;   // lanes per window, windows per tile, butterflies per lane and layer
    // (S < W: overlapping windows — each window's rows are loaded for it, the overlap comes out of the caches; HBM is read once)
    static constexpr bool ok = GeoT::kFixed && (base == 8 || base == 16) && layers >= 1 && (width == 16 || width == 64) && GeoT::S <= W && GeoT::S >= 1 && GeoT::D == 1 && GeoT::T == 0;
    static constexpr uint32_t kRows = TS / kSparkRow;                  // NCO rows per tile (2 or 4)
    static constexpr uint32_t lds_bytes(bool shift) { return ((W < 16u ? 16u : W) + 4u * TS) * 8u + (shift ? kSparkRow * 16u : 0u); }
};

template <int FMT, int NCO, class GeoT, int LB, int EPI /* the plan's qd_epilogue, compile-time: the tile loop carries no sink dispatch */>
__global__ __launch_bounds__(kThreads, LB) void k_spark2(const ChainParams P) {
    using K = Spark2<GeoT>;
    using FT = FmtTraits<FMT>;
    static_assert(K::ok, "k_spark2: W = base * 16 or base * 64");
    constexpr bool HAS_SHIFT = NCO != 0;
    constexpr uint32_t S = GeoT::S;                                         // samples between windows (== W: side by side)
    static_assert(S == K::W || !HAS_SHIFT, "k_spark2: overlapping windows only without a shift (the NCO rows are laid out for whole tiles)");
    constexpr uint32_t BPS = FT::BPS;                                      // a lane's column pair is 2 BPS bytes of a row: 16 (cf32), 8 (cs16) or 4 (cs8 / cu8)
    constexpr uint32_t W = K::W, logW = GeoT::logW, base = K::base, layers = K::layers, width = K::width, LPW = K::LPW, GW = K::GW, TS = K::TS, NBF = K::NBF;
    const GeoT geo(P);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *twl = reinterpret_cast<float2 *>(smem);                        // layer twiddles (W - base entries)
    double2 *jt = reinterpret_cast<double2 *>(twl + W);                    // NCO lane table (512 entries), chains with a shift only
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *fbw = twl + W + (HAS_SHIFT ? 2u * kSparkRow : 0u) + (size_t)wave * TS;
    for (uint32_t i = tid; i < W - base; i += kThreads) twl[i] = P.tw[i];
    if constexpr (HAS_SHIFT) { for (uint32_t i = tid; i < kSparkRow; i += kThreads) jt[i] = P.jtab[i]; }
    __syncthreads();                                                       // the only workgroup barrier

    // LDS swizzle of the transform buffer (Spark2Swz).  Unswizzled, the first layer (cols = base) reads point chunk * 4 base + k * base + i:
    // the 32 lanes of a half-wave span i and the low bits of `chunk`, which lie above bit 5 of the index — a 4-way (2-way) conflict on all of
    // that layer's reads and writes (rocprofv3, W = 128: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 54 %) —, and the base pass stores a
    // lane's butterfly outputs as 16-byte pieces of one contiguous run, piece q of 8 lanes per instruction group: with runs of 64 (128)
    // bytes a 4-way (8-way) conflict on every store of the pass.  The maps feed run- and chunk-index bits into index bits 1 ... 4 (whole
    // pieces, one XOR value per run); sigma is linear, so a position splits into a per-lane part (XOR value formed once) and compile-time
    // parts (folded), and with the buffers on 256-byte boundaries the XORs apply to the LDS byte address: one v_xor per access.
    // The bucket sink reads the buffer linearly afterwards: no swizzle there.
    constexpr bool kSwz = EPI != 2;
    using SZ = SparkSwzFn<Spark2Swz<kSwz ? W : 0u>>;
    const uint32_t fb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)fbw;      // LDS byte offset of the wave's buffer
    if (fb & 255u) __builtin_trap();                                       // (dynamic LDS starts at offset 0, every area before it is a multiple of 256 bytes: never taken)
    const uint32_t g = lane / LPW, xp = lane % LPW;                        // this lane's window of the tile and its column pair (2 xp, 2 xp + 1)
    // LDS byte addresses of the lane's two base butterflies' outputs (runs of `base` points; piece q at a ^ 16 q)
    const uint32_t r0 = g * W + (rev4(2 * xp, layers) << GeoT::log_base), r1 = g * W + (rev4(2 * xp + 1, layers) << GeoT::log_base);
    const uint32_t p0 = fb + ((r0 ^ SZ::delta(r0)) << 3), p1 = fb + ((r1 ^ SZ::delta(r1)) << 3);
    const uint64_t n_tiles = (P.n_windows + GW - 1) / GW;
    SparkWalk walk(n_tiles, wave);
    uint64_t tile = walk.first;
    const uint64_t n_waves = walk.stride, tile_end = walk.end;
    if (tile >= tile_end) return;
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto rsrc_of = [&](uint64_t t) {
        const uint64_t ns = (P.first_window + t * GW) * S, end = P.src_first + P.src_count;
        const uint64_t left = ns < end ? (end - ns) * BPS : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * BPS, 0,
                                                 left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
    };
    const uint32_t voff = (g * S + 2 * xp) * BPS;                           // lane's byte offset inside a row of its window
    // the integer formats arrive packed: a row's column pair is one dword (8-bit) or two (cs16), unpacked one column at a time in pass 1
    // — rows of 2 width bytes (8-bit, W = 128 / 256: 32 B per window and instruction; the row loads of a tile walk its cache lines in
    // order, each line is fetched from L2 once)
    using Raw = typename Spark2Raw<FMT>::type;
    Raw raw[base];
    auto load_rows = [&](const decltype(rsrc_of(0)) &rsrc) {
#pragma unroll
        for (uint32_t y = 0; y < base; ++y) {
            if constexpr (FMT == 0) raw[y] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(y * width * BPS), 2 /* nt */);
            else if constexpr (FMT == 3) raw[y] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(y * width * BPS), 2);
            else raw[y] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)(y * width * BPS), 2);
        }
    };
    auto column = [&](uint32_t y, int u) -> float2 {                        // sample (row y, column 2 xp + u) as the chain's unpack delivers it (src/lib.rs:241-255)
        if constexpr (FMT == 0) return u == 0 ? make_float2(__uint_as_float(raw[y].x), __uint_as_float(raw[y].y)) : make_float2(__uint_as_float(raw[y].z), __uint_as_float(raw[y].w));
        else if constexpr (FMT == 1) { const uint32_t a = raw[y] ^ 0x80808080u; return u == 0 ? make_float2(unpack_cs8_at(a, 0), unpack_cs8_at(a, 1)) : make_float2(unpack_cs8_at(a, 2), unpack_cs8_at(a, 3)); }
        else if constexpr (FMT == 2) return u == 0 ? make_float2(unpack_cu8_at(raw[y], 0), unpack_cu8_at(raw[y], 1)) : make_float2(unpack_cu8_at(raw[y], 2), unpack_cu8_at(raw[y], 3));
        else { const uint32_t w = u == 0 ? raw[y].x : raw[y].y; return make_float2(unpack_cs16(w & 0xffffu), unpack_cs16(w >> 16)); }
    };
    load_rows(rsrc_of(tile));
    if constexpr (EPI != 2) {
        // vmcnt retires in order and counts stores too.  Inside the loop the next tile's row loads are followed by this tile's 4 NBF
        // output stores before the rows are used at the top of the next iteration, so the wait there is vmcnt(4 NBF + base - 1 - y)
        // — but only if the loop's entry edge carries the same queue as its back edge: the waitcnt pass merges the two and, entering
        // with the loads alone, falls back to vmcnt(0), i.e. every tile waits for its own stores to land.  So the entry edge issues
        // the same number of stores through an EMPTY descriptor: the range check drops them, the counter sees them.
        const auto none = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out), 0, 0, 0x00020000);
#pragma unroll
        for (uint32_t q = 0; q < 4 * NBF; ++q) {
            if constexpr (EPI == 0) __builtin_amdgcn_raw_buffer_store_b32(0u, none, (int)(lane * 4 + q * 256), 0, 2);       // (distinct addresses: identical stores would be merged)
            else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, none, (int)(lane + q * 64), 0, 2);
        }
    }
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    while (true) {
        const uint64_t tile_n = tile + n_waves;
        const uint64_t w0 = P.first_window + tile * GW, left_w = P.first_window + P.n_windows - w0;
        const uint32_t g_cnt = left_w < GW ? (uint32_t)left_w : GW;
        constexpr uint32_t OBW = EPI == 0 ? 4u * W : W;                     // output bytes per window (norms f32 / glyph u8)
        const uint32_t RS = P.out_row_stride ? P.out_row_stride : 1u;      // rows between this launch's windows (interleaved launches of overlapping windows)
        const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out) + (w0 - P.out_window0) * RS * OBW, 0,
                                                             g_cnt ? ((g_cnt - 1) * RS + 1) * OBW : 0u, 0x00020000);
        // ---- NCO row bases of this lane: the tile's rows are wave-uniform (scalar loads), a lane needs the one(s) its window lies in
        RowBase rbl[W > kSparkRow ? W / kSparkRow : 1];
        if constexpr (HAS_SHIFT) {
            const_f64_p rows = (const_f64_p)(uintptr_t)(P.rowtab + (((w0 << logW) / kSparkRow) - P.rowtab_row0));
            constexpr uint32_t RPW = W > kSparkRow ? W / kSparkRow : 1;    // rows per window
#pragma unroll
            for (uint32_t h = 0; h < RPW; ++h) {
                // lane's row index inside the tile: windows of W >= 512 span RPW rows each, smaller ones share a row
                const uint32_t ridx = W >= kSparkRow ? g * RPW + h : (g * W) / kSparkRow;
                RowBase sel = load_rowbase_at(rows, 0);
#pragma unroll
                for (uint32_t r = 1; r < K::kRows; ++r) {
                    const RowBase cand = load_rowbase_at(rows, r);
                    if (ridx == r) sel = cand;
                }
                rbl[h] = sel;
            }
        }
        // ---- pass 1: the base butterflies of the lane's two columns, out of the row registers
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float2 v[base];
#pragma unroll
            for (uint32_t y = 0; y < base; ++y) {
                float2 x = column(y, u);
                if constexpr (HAS_SHIFT) {
                    const uint32_t jrow = (W >= kSparkRow ? 0u : (g * W) % kSparkRow) + (y * width) % kSparkRow;      // sample's place in its NCO row (before the column)
                    const uint32_t j = jrow + 2 * xp + (uint32_t)u;
                    const double2 cs = jt[j];
                    LaneRot lr; lr.jf = (double)j; lr.c = cs.x; lr.s = cs.y;
                    const RowBase &rb = rbl[W > kSparkRow ? (y * width) / kSparkRow : 0];
                    x = cmul_pk(x, nco_mul<NCO == 2>(rb, lr, P.ratio));     // buf[i] *= mul (src/shift.rs:51)
                }
                v[y] = x;
            }
#if !(defined(QD_SPARK_ABL) && (QD_SPARK_ABL & 8))
            if constexpr (base == 16) bf16(v, P.tw16_1, P.tw16_2, P.tw16_3, P.root2); else bf8(v, P.root2);
#endif
            const uint32_t pb = u == 0 ? p0 : p1;                          // byte address of the run's piece 0 in the swizzled layout
#pragma unroll
            for (uint32_t q = 0; q < base / 2; ++q)                         // piece q (outputs 2 q, 2 q + 1)
                *(spark_lds_f4 *)(uintptr_t)(pb ^ (16 * q)) = spark_f4n{v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y};
            __builtin_amdgcn_sched_barrier(0);                              // one column at a time (registers)
        }
        __builtin_amdgcn_sched_barrier(0);                                  // the rows are consumed: their registers take the next tile's rows
        load_rows(rsrc_of(tile_n < tile_end ? tile_n : tile));              // (last tile of this wave: harmless re-loads)
        __builtin_amdgcn_sched_barrier(0);
        // ---- radix-4 layers in place; the last one feeds the epilogue from registers
        uint32_t cols = base, log_cols = GeoT::log_base, tw_off = 0;
        uint32_t lo = lane;
        asm volatile("" : "+v"(lo));            // opaque per tile: the butterflies' LDS / output addresses are rebuilt, not hoisted out of the tile loop and spilled
#pragma unroll
        for (uint32_t l = 0; l < layers; ++l) {
            wsync();
#if defined(QD_SPARK_ABL) && (QD_SPARK_ABL & 4)
            if (l + 1 < layers) { tw_off += 3 * cols; cols *= 4; log_cols += 2; continue; }       // timing-only ablation: the last layer alone
#endif
            const bool last = l + 1 == layers;
            // layers of at most 64 columns: a lane's butterflies t = lane + 64 k share i = t mod cols, hence their three twiddles — read
            // once per tile and layer (cheaper than six registers per layer held across the whole tile loop, which spill)
            float2 tc1 = make_float2(0.f, 0.f), tc2 = tc1, tc3 = tc1;
            if (cols <= 64) { const uint32_t i = lo & (cols - 1); tc1 = twl[tw_off + 3 * i]; tc2 = twl[tw_off + 3 * i + 1]; tc3 = twl[tw_off + 3 * i + 2]; }
            // the lane's NBF butterflies two at a time: eight reads in flight, then the arithmetic; the scheduling barriers keep the
            // pairs (and, in the last layer, the four |X| of a butterfly) from being interleaved into one register-hungry block
            constexpr uint32_t KB = 2;
#pragma unroll
            for (uint32_t k0 = 0; k0 < NBF; k0 += KB) {
                float2 s[KB][4];
                uint32_t dp[KB][4];
#pragma unroll
                for (uint32_t kk = 0; kk < KB; ++kk) {
                    const uint32_t t = lo + 64 * (k0 + kk), chunk = t >> log_cols, i = t & (cols - 1);
                    const uint32_t pi = chunk * 4 * cols + i, b0 = fb + ((pi ^ SZ::delta(pi)) << 3);
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        // q cols (index bits log_cols, log_cols + 1, zero in pi): bits below 5 and the XOR value by XOR — inside the low 256 bytes,
                        // where the aligned base contributes nothing —, bits from 5 up by addition (an instruction offset)
                        const uint32_t qc = q * cols, lo_q = (qc ^ SZ::delta(qc)) & 31u, hi_q = qc & ~31u;
                        dp[kk][q] = (b0 ^ (lo_q << 3)) + (hi_q << 3);
                        s[kk][q] = spark_ld2(dp[kk][q]);
                    }
                }
#pragma unroll
                for (uint32_t kk = 0; kk < KB; ++kk) {
                    float2 t1, t2, t3;
                    if (cols <= 64) { t1 = tc1; t2 = tc2; t3 = tc3; }
                    else { const uint32_t i = (lo + 64 * (k0 + kk)) & (cols - 1); t1 = twl[tw_off + 3 * i]; t2 = twl[tw_off + 3 * i + 1]; t3 = twl[tw_off + 3 * i + 2]; }
                    s[kk][1] = cmul(s[kk][1], t1); s[kk][2] = cmul(s[kk][2], t2); s[kk][3] = cmul(s[kk][3], t3);
                    bf4(s[kk][0], s[kk][1], s[kk][2], s[kk][3]);
                }
                if (!last || EPI == 2) {
#pragma unroll
                    for (uint32_t kk = 0; kk < KB; ++kk)
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) spark_st2(dp[kk][q], s[kk][q]);
                } else {
                    // cols == W / 4: butterfly t of the tile is butterfly i of window t >> log_cols; result k is bin i + k W/4, output (bin + W/2) mod W.
                    // The four |X| of a butterfly take the short form together and test ONE flag for the IEEE form (3e-5 of the bins).
#pragma unroll
                    for (uint32_t kk = 0; kk < KB; ++kk) {
                        const uint32_t t = lo + 64 * (k0 + kk), gw = t >> log_cols, i = t & (cols - 1);
                        float nm[4];
                        bool sl[4];
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) {                  // output q * W/4 + i is bin ((q + 2) & 3) * W/4 + i
                            const float2 xv = s[kk][(q + 2) & 3];
#if defined(QD_SPARK_ABL) && (QD_SPARK_ABL & 2)
                            nm[q] = xv.x; sl[q] = false;                    // timing-only ablation: no |X|
#else
                            nm[q] = norm_fast(xv.x, xv.y, sl[q]);
#endif
                        }
                        if (__builtin_expect(sl[0] | sl[1] | sl[2] | sl[3], 0)) {
#pragma unroll
                            for (uint32_t q = 0; q < 4; ++q) if (sl[q]) nm[q] = norm_ieee(s[kk][(q + 2) & 3].x, s[kk][(q + 2) & 3].y);
                        }
#if defined(QD_SPARK_ABL) && (QD_SPARK_ABL & 1)
                        if (nm[0] + nm[1] + nm[2] + nm[3] != 12345.678f) continue;     // timing-only ablation: no output stores
#endif
                        // Stores through a per-tile buffer descriptor that ends with the tile's last VALID window: a store past it (the
                        // short last tile of a launch) is dropped by the range check, so the tile loop has no store branch, every tile
                        // issues the same number of vector-memory operations and the waits for the prefetched rows stay COUNTED
                        // (vmcnt retires in order: a conditional store between the row loads and their use forces vmcnt(0), i.e. a
                        // wait for the tile's own stores to land).  Non-temporal: the output is written once and not read here.
                        const uint32_t ob = ((gw * RS) << logW) + i;        // element offset inside the tile's output
                        if constexpr (EPI == 0) {
#pragma unroll
                            for (uint32_t q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nm[q]), orsrc, (int)((ob + q * cols) * 4), 0, 2);
                        } else {
#pragma unroll
                            for (uint32_t q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b8(glyph_of<GeoT>(P, nm[q]), orsrc, (int)(ob + q * cols), 0, 2);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            tw_off += 3 * cols; cols *= 4; log_cols += 2;
        }
        if constexpr (EPI == 2) {                                           // freq_levels: |X| through LDS, one lane per window sums the halves in order (src/fft.rs:95-97)
            wsync();
            wave_bucket_epilogue_fn<GeoT, TS / 64>(P, geo, fbw, w0, g_cnt, lane);
        }
        wsync();                                                            // the next tile's base outputs overwrite what this tile's layers read
        if (tile_n >= tile_end) break;
        tile = tile_n;
    }
}


// ---------------------------------------------------------------- k_spark0: overlapping windows of 2 ... 8 points straight out of registers (FLAGS bits 19 + 21)
//
// `from F sparkfft -width 4 -stride 2` — README example 1 and BASELINE configs[0]'s chain — at scale: the transform of such a window is
// ONE base butterfly (rustfft's Radix4 with no radix-4 layer: natural order in, natural order out), so a LANE owns a window from load to
// store: W BPS contiguous bytes at byte w S BPS of the stream (neighbouring lanes overlap; the coalescer and L1 serve that, HBM is read
// once), unpack, butterfly, |X| with the fftshift as a register renaming, and W norms (or glyph bytes, or the bucket digit) stored as one
// contiguous piece per lane — 64 lanes x 16 bytes = one 1-KiB store instruction for W = 4.  No LDS, no barrier; the next tile's loads go
// into the registers the unpack has just vacated.  Interleaved launches of k_spark (the stride dividing the width) read the stream W / S
// times and write rows W / S apart — half-filled store instructions: 2^28 cf32 samples at W = 4 / S = 2 took 2.42 ms there.
constexpr uint32_t kGeoSparkDirect = 2097152;   // FLAGS bit 21

template <int FMT, class GeoT, int LB, int EPI /* the plan's qd_epilogue */>
__global__ __launch_bounds__(kThreads, LB) void k_spark0(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    constexpr uint32_t W = GeoT::W, S = GeoT::S, BPS = FT::BPS, ND = W * BPS / 4;       // dwords per window
    static_assert(GeoT::kFixed && W >= 2 && W <= 16 && S >= 1 && S <= W && GeoT::D == 1 && GeoT::T == 0 && (W * BPS) % 4 == 0 && (S * BPS) % 4 == 0,
                  "k_spark0: one base butterfly per window, dword-aligned windows");
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_tiles = (P.n_windows + 63) / 64;                      // a tile: 64 windows, one per lane
    SparkWalk walk(n_tiles, wave);
    uint64_t tile = walk.first;
    const uint64_t n_waves = walk.stride, tile_end = walk.end;
    if (tile >= tile_end) return;
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto rsrc_of = [&](uint64_t t) {
        const uint64_t ns = (P.first_window + t * 64) * S, end = P.src_first + P.src_count;
        const uint64_t left = ns < end ? (end - ns) * BPS : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src) + (ns - P.src_first) * BPS, 0,
                                                 left > 0xffffffffull ? 0xffffffffu : (uint32_t)left, 0x00020000);
    };
    const uint32_t voff = lane * S * BPS;
    uint32_t raw[ND];
    auto load_window = [&](const decltype(rsrc_of(0)) &rsrc) {
#pragma unroll
        for (uint32_t d = 0; d < ND;) {
            if (ND - d >= 4) {
                const v4u_t w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(d * 4), 0);
                raw[d] = w.x; raw[d + 1] = w.y; raw[d + 2] = w.z; raw[d + 3] = w.w; d += 4;
            } else if (ND - d >= 2) {
                const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)(d * 4), 0);
                raw[d] = w.x; raw[d + 1] = w.y; d += 2;
            } else { raw[d] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)(d * 4), 0); d += 1; }
        }
    };
    auto sample = [&](uint32_t k) -> float2 {                              // src/lib.rs:241-255
        if constexpr (FMT == 0) return make_float2(__uint_as_float(raw[2 * k]), __uint_as_float(raw[2 * k + 1]));
        else if constexpr (FMT == 1) { const uint32_t a = raw[k / 2] ^ 0x80808080u; return make_float2(unpack_cs8_at(a, 2 * (k & 1)), unpack_cs8_at(a, 2 * (k & 1) + 1)); }
        else if constexpr (FMT == 2) return make_float2(unpack_cu8_at(raw[k / 2], 2 * (k & 1)), unpack_cu8_at(raw[k / 2], 2 * (k & 1) + 1));
        else return make_float2(unpack_cs16(raw[k] & 0xffffu), unpack_cs16(raw[k] >> 16));
    };
    constexpr uint32_t OBW = EPI == 0 ? 4u * W : (EPI == 1 ? W : 1u);     // output bytes per window
    constexpr uint32_t NST = EPI == 0 ? (W >= 4 ? W / 4 : 1u) : 1u;        // store instructions per tile
    load_window(rsrc_of(tile));
    {
        // the loop's entry edge issues as many (dropped) stores as a tile does: the waits for the prefetched windows stay counted (see k_spark2)
        const auto none = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out), 0, 0, 0x00020000);
#pragma unroll
        for (uint32_t q = 0; q < NST; ++q) __builtin_amdgcn_raw_buffer_store_b32(0u, none, (int)(lane * 4 + q * 256), 0, 2);
    }
    while (true) {
        const uint64_t tile_n = tile + n_waves;
        const uint64_t w0 = P.first_window + tile * 64, left_w = P.first_window + P.n_windows - w0;
        const uint32_t g_cnt = left_w < 64 ? (uint32_t)left_w : 64u;
        const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(P.out) + (w0 - P.out_window0) * OBW, 0, g_cnt * OBW, 0x00020000);
        float2 v[W];
#pragma unroll
        for (uint32_t k = 0; k < W; ++k) v[k] = sample(k);
        __builtin_amdgcn_sched_barrier(0);                                  // the window is unpacked: its registers take the next tile's
        load_window(rsrc_of(tile_n < tile_end ? tile_n : tile));            // (last tile of this wave: harmless re-loads)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (W == 16) bf16(v, P.tw16_1, P.tw16_2, P.tw16_3, P.root2);
        else if constexpr (W == 8) bf8(v, P.root2);
        else if constexpr (W == 4) bf4(v[0], v[1], v[2], v[3]);
        else bf2(v[0], v[1]);
        float nm[W];                                                        // nm[o] = |X[(o + W/2) mod W]| (src/fft.rs:48-52; bucket sink: |X[o]|)
#pragma unroll
        for (uint32_t o0 = 0; o0 < W; o0 += 4) {
            constexpr uint32_t NQ = W >= 4 ? 4 : W;
            bool sl[NQ];
            bool any = false;
            constexpr uint32_t kShift = EPI == 2 ? 0u : W / 2;             // (freq_levels sums the bins in their own order: src/fft.rs:86-97)
#pragma unroll
            for (uint32_t q = 0; q < NQ; ++q) { const float2 x = v[(o0 + q) ^ kShift]; nm[o0 + q] = norm_fast(x.x, x.y, sl[q]); any |= sl[q]; }
            if (__builtin_expect(any, 0)) {
#pragma unroll
                for (uint32_t q = 0; q < NQ; ++q) if (sl[q]) { const float2 x = v[(o0 + q) ^ kShift]; nm[o0 + q] = norm_ieee(x.x, x.y); }
            }
        }
        if constexpr (EPI == 0) {
            if constexpr (W >= 4) {
#pragma unroll
                for (uint32_t q = 0; q < W / 4; ++q) {
                    const v4u_t o = {__float_as_uint(nm[4 * q]), __float_as_uint(nm[4 * q + 1]), __float_as_uint(nm[4 * q + 2]), __float_as_uint(nm[4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)(lane * OBW), (int)(q * 16), 2);
                }
            } else {
                const v2u_t o = {__float_as_uint(nm[0]), __float_as_uint(nm[1])};
                __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, (int)(lane * OBW), 0, 2);
            }
        } else if constexpr (EPI == 1) {
            uint32_t pk[(W + 3) / 4] = {};
#pragma unroll
            for (uint32_t o = 0; o < W; ++o) pk[o / 4] |= (uint32_t)glyph_of<GeoT>(P, nm[o]) << (8 * (o & 3));
            if constexpr (W == 16) { const v4u_t o = {pk[0], pk[1], pk[2], pk[3]}; __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)(lane * OBW), 0, 2); }
            else if constexpr (W == 8) { const v2u_t o = {pk[0], pk[1]}; __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, (int)(lane * OBW), 0, 2); }
            else if constexpr (W == 4) __builtin_amdgcn_raw_buffer_store_b32(pk[0], orsrc, (int)(lane * OBW), 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pk[0], orsrc, (int)(lane * OBW), 0, 2);
        } else {
            float first = 0.f, second = 0.f;                                // src/fft.rs:95-97: the halves summed in order
#pragma unroll
            for (uint32_t k = 0; k < W / 2; ++k) first = first + nm[k];
#pragma unroll
            for (uint32_t k = W / 2; k < W; ++k) second = second + nm[k];
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(first < second ? 0 : 1), orsrc, (int)lane, 0, 2);
        }
        if (tile_n >= tile_end) break;
        tile = tile_n;
    }
}


}  // namespace qd
