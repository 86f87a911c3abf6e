// qd_chain.h — the fused chain kernel:  unpack -> shift -> lowpass(FIR+decimate) -> FFT -> |X| -> epilogue
//
// One workgroup (256 threads) owns a *tile* of G consecutive FFT windows of the sink's loop
// (spark_fft, src/fft.rs:28-65).  Per tile:
//   phase 1  stream the tile's contiguous raw range from HBM (16-byte coalesced loads), unpack,
//            multiply by the NCO, park the shifted cf32 samples in LDS (row-padded so that the
//            FIR's stride-D lane pattern is bank-conflict free);
//   phase 2  FIR+decimate, one lane per decimated output, taps ascending, separately rounded
//            mul/add — the reference's summation order (src/filter.rs:111-121), including its
//            per-read_at tail truncation (jmax), results scattered into the FFT buffer in
//            rustfft's digit-reversed order;
//   phase 3  W-point FFT in LDS with rustfft's scalar Radix4 structure (base butterfly +
//            radix-4 DIT layers);
//   phase 4  fftshift + hypot (+ glyph / bucket) and a coalesced store.
// Decimated samples never leave the CU.  Windows are independent units, so tiles need no
// inter-workgroup communication; the grid is sized to the chip and strides over tiles.
#pragma once

#include "qd_device.h"

namespace qd {

constexpr int kThreads = 256;

// Wave-uniform read-only tables (taps, row bases) are read through the constant address space
// so that hipcc emits scalar loads (s_load_dwordx8) instead of one vector load per lane.
typedef const float __attribute__((address_space(4))) *const_f32_p;
typedef const double __attribute__((address_space(4))) *const_f64_p;

// samples per lane per row-load, by format
template <int FMT> struct FmtTraits;
template <> struct FmtTraits<0> { static constexpr int BPS = 8; static constexpr int SPL = 2; };  // cf32: 16 B / lane
template <> struct FmtTraits<1> { static constexpr int BPS = 2; static constexpr int SPL = 4; };  // cs8 :  8 B / lane
template <> struct FmtTraits<2> { static constexpr int BPS = 2; static constexpr int SPL = 4; };  // cu8 :  8 B / lane
template <> struct FmtTraits<3> { static constexpr int BPS = 4; static constexpr int SPL = 4; };  // cs16: 16 B / lane

struct ChainParams {
    const uint8_t *src;        // raw bytes of sample src_first
    uint64_t src_first;
    uint64_t src_count;
    uint64_t first_window;     // of this launch
    uint64_t n_windows;
    uint64_t out_window0;      // window index that maps to out[0]
    uint32_t W, logW, S, D, T, c, G;
    uint32_t Dp;               // LDS row pitch: D + 1 if D even else D
    uint32_t dmagic;           // floor(2^32 / D) + 1  (exact m / D for m*D < 2^32)
    uint32_t a0, b0;           // c = a0*D + b0
    uint32_t T_fast;           // min(T, D + T/2): every output's jmax is >= this
    uint32_t a1, b1;           // c + T_fast = a1*D + b1
    uint32_t base_len, log_base, layers; // rustfft Radix4 plan: W = base_len * 4^layers
    uint32_t vec_ok;           // vector loads are aligned
    uint32_t second_order;     // NCO second-order correction
    uint32_t epi;              // qd_epilogue
    uint32_t lds_raw_elems;    // float2 capacity of the raw tile
    float rmin, rmax;
    float root2;
    float2 tw16_1, tw16_2, tw16_3;
    double ratio;
    const RowBase *rowtab;     // indexed by absolute row - rowtab_row0
    uint64_t rowtab_row0;
    const double2 *jtab;       // ROW entries (cos, sin)(fl(j*ratio))
    const float *taps;
    const float2 *tw;          // radix-4 layer twiddles, bottom layer first
    void *out;
};

template <int FMT>
__device__ __forceinline__ float2 load_sample_scalar(const uint8_t *src, uint64_t idx, const float *lut) {
    if constexpr (FMT == 0) {
        return *reinterpret_cast<const float2 *>(src + idx * 8);
    } else if constexpr (FMT == 1 || FMT == 2) {
        uint16_t v = *reinterpret_cast<const uint16_t *>(src + idx * 2);
        return make_float2(lut[v & 0xff], lut[v >> 8]);
    } else {
        uint32_t v = *reinterpret_cast<const uint32_t *>(src + idx * 4);
        return make_float2(unpack_cs16(v & 0xffffu), unpack_cs16(v >> 16));
    }
}

// FIR over taps [j0, j1) for one output; (rowp, b) is the LDS position of tap j0.
// Control flow is wave-uniform; only rowp and jmax differ per lane.
template <bool PRED>
__device__ __forceinline__ void fir_span(const float2 *rowp, uint32_t b, uint32_t j0, uint32_t j1, uint32_t jmax,
                                         uint32_t D, uint32_t Dp, const float *__restrict__ taps,
                                         float &accr, float &acci) {
    uint32_t j = j0;
    while (j < j1) {
        uint32_t run = D - b;
        if (run > j1 - j) run = j1 - j;
        const float2 *p = rowp + b;
        const_f32_p h = (const_f32_p)(uintptr_t)(taps + j);
#pragma unroll 8
        for (uint32_t i = 0; i < run; ++i) {
            float2 x = p[i];
            float hh = h[i];
            if (!PRED || (j + i) < jmax) {
                accr = accr + x.x * hh;   // Complex<f32> * f32, then +=  (src/filter.rs:119)
                acci = acci + x.y * hh;
            }
        }
        j += run;
        b = 0;
        rowp += Dp;
    }
}

template <int FMT, bool HAS_SHIFT, bool HAS_FIR>
__global__ __launch_bounds__(kThreads) void k_chain(const ChainParams P) {
    using FT = FmtTraits<FMT>;
    constexpr int SPL = FT::SPL;
    constexpr int BPS = FT::BPS;
    constexpr uint32_t ROW = kThreads * SPL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *raw = reinterpret_cast<float2 *>(smem);
    float2 *fb = raw + P.lds_raw_elems;
    float *lut = reinterpret_cast<float *>(fb + (size_t)P.G * P.W);

    const uint32_t tid = threadIdx.x;
    const uint32_t W = P.W, logW = P.logW, S = P.S, D = P.D, T = P.T, Dp = P.Dp;

    if constexpr (FMT == 1) lut[tid] = unpack_cs8(tid);
    if constexpr (FMT == 2) lut[tid] = unpack_cu8(tid);

    LaneRot lr[SPL];
    if constexpr (HAS_SHIFT) {
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            uint32_t j = tid * SPL + u;
            double2 cs = P.jtab[j];
            lr[u].jf = (double)j;
            lr[u].tj = lr[u].jf * P.ratio;
            lr[u].c = cs.x;
            lr[u].s = cs.y;
        }
    }
    if constexpr (FMT == 1 || FMT == 2) __syncthreads();

    const uint64_t n_tiles = (P.n_windows + P.G - 1) / P.G;
    const uint64_t src_end = P.src_first + P.src_count;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = P.first_window + tile * P.G;
        uint64_t left = P.first_window + P.n_windows - w0;
        const uint32_t g_cnt = left < P.G ? (uint32_t)left : P.G;
        const uint64_t n_start = w0 * S * D;                       // LowPass reads inner at off*D (src/filter.rs:71)
        const uint32_t tile_raw = (g_cnt - 1) * S * D + W * D + T; // B*D + T per window (src/filter.rs:68)
        const uint64_t n_end = n_start + tile_raw;

        // ---------------- phase 1: HBM -> unpack -> NCO -> LDS
        const uint64_t r0 = n_start / ROW, r1 = (n_end + ROW - 1) / ROW;
        for (uint64_t r = r0; r < r1; ++r) {
            const uint64_t nrow = r * ROW;
            const uint64_t n0 = nrow + (uint64_t)tid * SPL;
            // wave-uniform skip of waves entirely outside the tile
            const uint64_t wv0 = nrow + (uint64_t)(tid & ~63u) * SPL;
            if (wv0 >= n_end || wv0 + 64 * SPL <= n_start) continue;
            const bool any = (n0 + SPL > n_start) && (n0 < n_end);
            if (!any) continue;

            float2 x[SPL];
            const bool whole = P.vec_ok && n0 >= P.src_first && n0 + SPL <= src_end;
            if (whole) {
                const uint8_t *p = P.src + (n0 - P.src_first) * BPS;
                if constexpr (FMT == 0) {
                    float4 v = *reinterpret_cast<const float4 *>(p);
                    x[0] = make_float2(v.x, v.y);
                    x[1] = make_float2(v.z, v.w);
                } else if constexpr (FMT == 1 || FMT == 2) {
                    uint2 v = *reinterpret_cast<const uint2 *>(p);
                    x[0] = make_float2(lut[v.x & 0xff], lut[(v.x >> 8) & 0xff]);
                    x[1] = make_float2(lut[(v.x >> 16) & 0xff], lut[v.x >> 24]);
                    x[2] = make_float2(lut[v.y & 0xff], lut[(v.y >> 8) & 0xff]);
                    x[3] = make_float2(lut[(v.y >> 16) & 0xff], lut[v.y >> 24]);
                } else {
                    uint4 v = *reinterpret_cast<const uint4 *>(p);
                    x[0] = make_float2(unpack_cs16(v.x & 0xffffu), unpack_cs16(v.x >> 16));
                    x[1] = make_float2(unpack_cs16(v.y & 0xffffu), unpack_cs16(v.y >> 16));
                    x[2] = make_float2(unpack_cs16(v.z & 0xffffu), unpack_cs16(v.z >> 16));
                    x[3] = make_float2(unpack_cs16(v.w & 0xffffu), unpack_cs16(v.w >> 16));
                }
            } else {
#pragma unroll
                for (int u = 0; u < SPL; ++u) {
                    uint64_t n = n0 + u;
                    x[u] = (n >= n_start && n < n_end) ? load_sample_scalar<FMT>(P.src, n - P.src_first, lut)
                                                        : make_float2(0.f, 0.f);
                }
            }

            if constexpr (HAS_SHIFT) {
                const_f64_p rp = (const_f64_p)(uintptr_t)(P.rowtab + (r - P.rowtab_row0));   // uniform: scalar loads
                RowBase rb;
                rb.c = rp[0]; rb.s = rp[1]; rb.theta = rp[2]; rb.nf = rp[3];
#pragma unroll
                for (int u = 0; u < SPL; ++u) {
                    float2 m = nco_mul(rb, lr[u], P.ratio, P.second_order != 0);
                    x[u] = cmul(x[u], m);                          // buf[i] *= mul (src/shift.rs:51)
                }
            }

            // tile-relative index -> padded LDS position  m + (m / D) * (Dp - D)
            int64_t mrel = (int64_t)(n0 - n_start);
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                int64_t m = mrel + u;
                if (m >= 0 && m < (int64_t)tile_raw) {
                    uint32_t mu = (uint32_t)m;
                    uint32_t q = __umulhi(mu, P.dmagic);
                    raw[mu + q * (Dp - D)] = x[u];
                }
            }
        }
        __syncthreads();

        // ---------------- phase 2: FIR + decimate (or plain window gather), scatter for the FFT
        const uint32_t n_out = g_cnt << logW;
        const uint32_t log_width = 2 * P.layers;  // width = W / base_len = 4^layers
        for (uint32_t o = tid; o < n_out; o += kThreads) {
            const uint32_t g = o >> logW, k = o & (W - 1);
            const uint32_t q = g * S + k;
            float accr = 0.f, acci = 0.f;
            if constexpr (HAS_FIR) {
                // jmax(k) = min(T, valid - (k*D + c)) with valid = W*D + T (full read)
                uint32_t jmax = (W - k) * D + T / 2;
                if (jmax > T) jmax = T;
                const float2 *rowp = raw + (size_t)(q + P.a0) * Dp;
                const bool wave_full = __all(jmax == T);
                if (wave_full) {
                    fir_span<false>(rowp, P.b0, 0, T, T, D, Dp, P.taps, accr, acci);
                } else {
                    fir_span<false>(rowp, P.b0, 0, P.T_fast, T, D, Dp, P.taps, accr, acci);
                    const float2 *rowp1 = raw + (size_t)(q + P.a1) * Dp;
                    fir_span<true>(rowp1, P.b1, P.T_fast, T, jmax, D, Dp, P.taps, accr, acci);
                }
            } else {
                float2 v = raw[q];
                accr = v.x; acci = v.y;
            }
            // bitreversed_transpose::<4>(base_len, ..): out[y + rev(x)*base] = in[x + y*width]
            const uint32_t xx = k & ((1u << log_width) - 1), yy = k >> log_width;
            const uint32_t pos = yy + rev4(xx, P.layers) * P.base_len;
            fb[(g << logW) + pos] = make_float2(accr, acci);
        }
        __syncthreads();

        // ---------------- phase 3: FFT (rustfft Radix4: base butterflies, then radix-4 layers)
        {
            const uint32_t base = P.base_len;
            const uint32_t log_tpw = logW - P.log_base;   // base tasks per window = W / base
            const uint32_t n_task = g_cnt << log_tpw;
            for (uint32_t t = tid; t < n_task; t += kThreads) {
                // task t -> window t >> log_tpw, chunk t & (tpw-1): chunks are contiguous, so
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
            uint32_t cols = base, log_cols = P.log_base;
            const float2 *tw = P.tw;
            for (uint32_t layer = 0; layer < P.layers; ++layer) {
                __syncthreads();
                const uint32_t n_bf = (g_cnt << logW) >> 2;   // W/4 butterflies per window
                for (uint32_t t = tid; t < n_bf; t += kThreads) {
                    // butterfly t: chunk (of 4*cols, windows are contiguous) t >> log_cols, column t & (cols-1)
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
        __syncthreads();

        // ---------------- phase 4: fftshift + norm + epilogue
        const uint64_t wrel = w0 - P.out_window0;
        if (P.epi == 2) {
            // freq_levels (src/fft.rs:95-97): sequential f32 sums of |X[k]| over each half
            float *nb = reinterpret_cast<float *>(raw);       // raw tile is dead now
            for (uint32_t o = tid; o < n_out; o += kThreads) nb[o] = norm_ref(fb[o]);
            __syncthreads();
            if (tid < g_cnt) {
                const float *p = nb + (tid << logW);
                float first = 0.f, second = 0.f;
                for (uint32_t k = 0; k < W / 2; ++k) first = first + p[k];
                for (uint32_t k = W / 2; k < W; ++k) second = second + p[k];
                reinterpret_cast<uint8_t *>(P.out)[wrel + tid] = first < second ? 0 : 1;
            }
        } else {
            for (uint32_t o = tid; o < n_out; o += kThreads) {
                const uint32_t g = o >> logW, b = o & (W - 1);
                const float nm = norm_ref(fb[(g << logW) + ((b + W / 2) & (W - 1))]);
                const uint64_t oi = ((wrel + g) << logW) + b;
                if (P.epi == 0) reinterpret_cast<float *>(P.out)[oi] = nm;
                else reinterpret_cast<uint8_t *>(P.out)[oi] = glyph_code(nm, P.rmin, P.rmax);
            }
        }
        __syncthreads();
    }
}

}  // namespace qd
