// quadrs_hip.hip — C ABI (include/quadrs_hip.h) + host-side planning for the gfx950 engine.
//
// Host responsibilities (all O(plan), none per-sample): validate the chain the way the
// reference's constructors do, design the taps with the platform libm (src/filter.rs:86-105 —
// the reference does this on the host too), lay out twiddles, build the NCO tables on the
// device, pick the tile geometry, launch.  Per-sample work lives in qd_chain.h / qd_device.h.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <map>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <algorithm>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/quadrs_hip.h"
#include "qd_chain.h"
#include "qd_registry.h"

using namespace qd;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) return fail(QD_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e__)); \
    } while (0)

constexpr double kPi64 = 3.14159265358979323846264338327950288;
constexpr float kPi32 = 3.14159265358979323846264338327950288f;

// Tuning / ablation knobs read from the environment exist in development builds only (-DQD_DEVELOP:
// libquadrs_hip_dev.so, used by scripts/); the shipped library takes its policy from qd_plan_options.
#ifdef QD_DEVELOP
const char *dev_env(const char *name) { return getenv(name); }
#else
const char *dev_env(const char *) { return nullptr; }
#endif

uint32_t ilog2(uint64_t v) { uint32_t l = 0; while ((1ull << l) < v) ++l; return l; }
bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }

int spl_of(int fmt) { return fmt == QD_FMT_CF32 ? 2 : 4; }
int bps_of(int fmt) { return fmt == QD_FMT_CF32 ? 8 : (fmt == QD_FMT_CS16 ? 4 : 2); }

// ------------------------------------------------------------------ small kernels

// Row bases: (cos, sin) of the exact product (row*ROW) * ratio (qd_device.h, NCO), plus nf itself.
// (n_off: the row grid of a stream that starts n_off samples into the plan's — the interleaved launches of overlapping lowpass-free windows)
__global__ void k_rowtab(double ratio, uint32_t row_len, uint64_t row0, uint64_t n_rows, uint64_t n_off, RowBase *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    RowBase rb;
    rb.nf = (double)((row0 + i) * (uint64_t)row_len + n_off);
    rb.pad_ = 0.0;
    nco_table_entry(rb.nf, ratio, &rb.c, &rb.s);
    out[i] = rb;
}

// Lane table: (cos, sin) of the exact product j * ratio
__global__ void k_jtab(double ratio, uint32_t n, double2 *out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s, c;
    nco_table_entry((double)j, ratio, &c, &s);
    out[j] = make_double2(c, s);
}

// FileFormat::to_cf32 per sample (src/lib.rs:231-255)
__global__ void k_unpack(int fmt, const uint8_t *src, size_t n, float2 *out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float2 v;
        switch (fmt) {
        case 0: v = *reinterpret_cast<const float2 *>(src + i * 8); break;
        case 1: { uint16_t w = *reinterpret_cast<const uint16_t *>(src + i * 2); v = make_float2(unpack_cs8(w & 0xff), unpack_cs8(w >> 8)); break; }
        case 2: { uint16_t w = *reinterpret_cast<const uint16_t *>(src + i * 2); v = make_float2(unpack_cu8(w & 0xff), unpack_cu8(w >> 8)); break; }
        default: { uint32_t w = *reinterpret_cast<const uint32_t *>(src + i * 4); v = make_float2(unpack_cs16(w & 0xffffu), unpack_cs16(w >> 16)); break; }
        }
        out[i] = v;
    }
}

// Shift::read_at's loop over an arbitrary block (src/shift.rs:48-52), rows of 512 samples.
__global__ __launch_bounds__(256) void k_shift(float2 *buf, uint64_t abs_off, uint64_t n, double ratio,
                                                const RowBase *rowtab, uint64_t row0, uint64_t n_rows,
                                                const double2 *jtab, int second_order) {
    constexpr uint32_t ROW = 512;
    const uint32_t tid = threadIdx.x;
    LaneRot lr[2];
    for (int u = 0; u < 2; ++u) {
        uint32_t j = tid * 2 + u;
        double2 cs = jtab[j];
        lr[u].jf = (double)j; lr[u].c = cs.x; lr[u].s = cs.y;
    }
    for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const RowBase rb = rowtab[r];
        for (int u = 0; u < 2; ++u) {
            uint64_t idx = (row0 + r) * ROW + tid * 2 + u;
            if (idx >= abs_off && idx < abs_off + n) {
                float2 m = second_order ? nco_mul<true>(rb, lr[u], ratio) : nco_mul<false>(rb, lr[u], ratio);
                buf[idx - abs_off] = cmul(buf[idx - abs_off], m);
            }
        }
    }
}

// LowPass::read_at on a fetched block (src/filter.rs:68-83,107-124): one lane per kept output.
__global__ void k_lowpass_block(const float *__restrict__ taps, uint32_t T, uint64_t D, const float2 *__restrict__ raw,
                                uint64_t valid, float2 *out, uint64_t out_n) {
    const uint64_t c = T - T / 2;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < out_n; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t base = k * D + c;
        uint64_t jmax = valid - base < T ? valid - base : T;
        float ar = 0.f, ai = 0.f;
        for (uint64_t j = 0; j < jmax; ++j) {
            float2 x = raw[base + j];
            float h = taps[j];
            ar = ar + x.x * h;
            ai = ai + x.y * h;
        }
        out[k] = make_float2(ar, ai);
    }
}

// Gen::read_at (src/gen.rs:35-47)
__global__ void k_gen(const int64_t *cos_hz, uint32_t n_cos, uint64_t sample_rate, uint64_t first, size_t n, float2 *out) {
    const double tau = kPi64 * 2.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double base = (double)(first + i) * tau / (double)sample_rate;
        float vr = 0.f, vi = 0.f;
        for (uint32_t k = 0; k < n_cos; ++k) {
            double f = (double)cos_hz[k] * base;
            double s, c;
            sincos(f, &s, &c);
            vr = vr + (float)c;
            vi = vi + (float)s;
        }
        out[i] = make_float2(vr, vi);
    }
}

// take_fft at a width that is not a power of two (src/ffts.rs:25: FftPlanner::plan_fft_forward takes any length; the
// egui slider offers 4..4096, src/eui/mod.rs:157).  Bluestein's chirp-z form: X[k] = c[k] * sum_n (x[n] c[n]) b[k-n],
// c[n] = e^{-i pi n^2 / W}, b = conj(c) — a circular convolution of length M >= 2W-1 (a power of two), done in LDS as
// forward radix-2 DIF (natural in, bit-reversed out) -> pointwise product with the precomputed spectrum of b (stored
// bit-reversed, 1/M folded in) -> inverse radix-2 DIT (bit-reversed in, natural out): no permutation pass.
// One workgroup per output row.  rustfft's own result for such lengths depends on the planner's decomposition and the
// host's SIMD code path, so there is no bit pattern to match (PARITY UNPINNED).  The convolution is therefore carried in
// f64 (the f32 window product of src/ffts.rs:64-68 first, exactly as the reference forms it): the bins come out as the
// mathematically exact DFT of the windowed f32 samples rounded once to f32 — an f32 Bluestein would sit 5-10x further from
// the exact answer than rustfft's mixed-radix paths do on smooth lengths.  Cost: ~1 ms for 2048 rows at M = 8192.
__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__global__ __launch_bounds__(256) void k_bluestein(const float2 *__restrict__ in, uint64_t in_first, const uint64_t *__restrict__ offs,
                                                   const float *__restrict__ win, uint32_t W, uint32_t M, uint32_t logM,
                                                   const double2 *__restrict__ chirp, const double2 *__restrict__ Bbr,
                                                   const double2 *__restrict__ tw, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    double2 *buf = reinterpret_cast<double2 *>(smem_b);
    const uint32_t tid = threadIdx.x;
    const uint64_t row = blockIdx.x;
    const float2 *x = in + (offs[row] - in_first);
    for (uint32_t n = tid; n < M; n += 256) {
        double2 v = make_double2(0.0, 0.0);
        if (n < W) {
            float2 xv = x[n];
            if (win) xv = cscale(xv, win[n]);         // *sample *= w_val (src/ffts.rs:64-68), Complex<f32> * f32, f32-rounded
            v = zmul(make_double2((double)xv.x, (double)xv.y), chirp[n]);
        }
        buf[n] = v;
    }
    __syncthreads();
    for (uint32_t sh = logM; sh-- > 0;) {             // DIF, half-span h = 2^sh
        const uint32_t h = 1u << sh;
        for (uint32_t t = tid; t < M / 2; t += 256) {
            const uint32_t j = t & (h - 1), i = ((t >> sh) << (sh + 1)) | j;
            const double2 u = buf[i], v = buf[i + h];
            buf[i] = make_double2(u.x + v.x, u.y + v.y);
            buf[i + h] = zmul(make_double2(u.x - v.x, u.y - v.y), tw[j << (logM - 1 - sh)]);
        }
        __syncthreads();
    }
    for (uint32_t r = tid; r < M; r += 256) buf[r] = zmul(buf[r], Bbr[r]);
    __syncthreads();
    for (uint32_t sh = 0; sh < logM; ++sh) {          // DIT with conjugate twiddles
        const uint32_t h = 1u << sh;
        for (uint32_t t = tid; t < M / 2; t += 256) {
            const uint32_t j = t & (h - 1), i = ((t >> sh) << (sh + 1)) | j;
            const double2 w = tw[j << (logM - 1 - sh)];
            const double2 u = buf[i], v = zmul(buf[i + h], make_double2(w.x, -w.y));
            buf[i] = make_double2(u.x + v.x, u.y + v.y);
            buf[i + h] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
    const uint32_t half = W / 2;                       // skip(W/2).chain(take(W/2)), src/ffts.rs:72-78
    for (uint32_t k = tid; k < W; k += 256) {
        const double2 y = zmul(buf[k], chirp[k]);
        const uint32_t pos = k >= half ? k - half : k + (W - half);
        out[row * W + pos] = norm_ref(make_float2((float)y.x, (float)y.y));     // the FFT result is Complex<f32>; norm() = hypotf
    }
}

// ------------------------------------------------------------------ host arithmetic restated from the reference

// src/filter.rs:86-105 with cutoff from :126-128,:31 — f32 throughout, platform libm
void design_taps(uint64_t frequency, uint64_t sample_rate, size_t size, float *out) {
    float cutoff = (float)((double)frequency / (double)sample_rate);
    float sz1 = (float)size - 1.0f;
    for (size_t i = 0; i < size; ++i) {
        float fi = (float)i;
        float a1 = (2.0f * kPi32) * fi / sz1;
        float a2 = (4.0f * kPi32) * fi / sz1;
        float window = 0.42f - 0.5f * std::cos(a1) + 0.08f * std::cos(a2);
        float x = 2.0f * cutoff * (fi - sz1 / 2.0f);
        float xp = x * kPi32;
        float wave = std::sin(xp) / xp;
        out[i] = wave * window;
    }
    float sum = 0.0f;
    for (size_t i = 0; i < size; ++i) sum += out[i];
    for (size_t i = 0; i < size; ++i) out[i] = out[i] / sum;
}

// rustfft twiddles::compute_twiddle, Forward
float2 compute_twiddle(size_t index, size_t fft_len) {
    double constant = -2.0 * kPi64 / (double)fft_len;
    double angle = constant * (double)index;
    return make_float2((float)std::cos(angle), (float)std::sin(angle));
}

struct FftLayout {
    uint32_t base_len = 1, log_base = 0, layers = 0;
    std::vector<float2> tw;
};

// Radix4::new: exponent 0..3 -> base 1,2,4,8; else odd -> 8, even -> 16
FftLayout fft_layout(uint64_t W) {
    FftLayout L;
    uint32_t e = ilog2(W);
    uint32_t be = e <= 3 ? e : ((e & 1) ? 3 : 4);
    L.log_base = be; L.base_len = 1u << be; L.layers = (e - be) / 2;
    size_t cross = L.base_len;
    while (cross < W) {
        size_t cols = cross;
        cross *= 4;
        for (size_t i = 0; i < cols; ++i)
            for (size_t k = 1; k < 4; ++k) L.tw.push_back(compute_twiddle(i * k, cross));
    }
    return L;
}


#ifdef QD_DEV_FAST   // development builds: cf32 only, to keep hipcc turnaround short
#define QD_FMT_CASES(X) case 0: return X(0);
#else
#define QD_FMT_CASES(X) case 0: return X(0); case 1: return X(1); case 2: return X(2); case 3: return X(3);
#endif

// ---- generic kernels (DynGeo): every shape; chunked prefetch of 4 rows; aligned / unaligned slab
// Register budget of the runtime-geometry kernels: four waves per SIMD (128 VGPRs).  Round 3's builds spilled there once a shift was in
// the chain (up to 74 VGPRs / 108 bytes of scratch per lane on the cs16 and second-order instantiations, reloaded behind vmcnt(0)
// drains in the tile loop).  Round 4: the lane constants are re-read from the (L2-resident) lane table at the top of every tile
// instead of held across it (k_chain kReloadLane) — no scratch at four waves for the first-order NCO; the second-order kernels (streams
// past 2^28 rad of phase, which get a plan-time build anyway) are budgeted for three.  Budgeting ALL shifted kernels for three waves
// removed the scratch too but cost 6-18 % on 256 MiB streams (profiles/r04/generic_rate.log).  tests/test_abi_cpu.py audits every kernel.
constexpr int dyn_lb(int nco) { return nco == 2 ? 3 : 4; }
template <int F, int NCO, bool FI>
chain_fn pick_dyn(bool aligned) {
    return aligned ? k_chain<F, NCO, DynGeo, FI, 4, false, true, dyn_lb(NCO)> : k_chain<F, NCO, DynGeo, FI, 4, false, false, dyn_lb(NCO)>;
}

template <int F>
chain_fn pick_fmt(int nco, bool fir, bool aligned) {
    switch (nco) {
    case 0: return fir ? pick_dyn<F, 0, true>(aligned) : pick_dyn<F, 0, false>(aligned);
    case 1: return fir ? pick_dyn<F, 1, true>(aligned) : pick_dyn<F, 1, false>(aligned);
    default: return fir ? pick_dyn<F, 2, true>(aligned) : pick_dyn<F, 2, false>(aligned);
    }
}

chain_fn pick_generic(int fmt, int nco, bool fir, bool aligned) {
#define QD_X(F) pick_fmt<F>(nco, fir, aligned)
    switch (fmt) { QD_FMT_CASES(QD_X) }
#undef QD_X
    return nullptr;
}

// ---- chains without a lowpass, windows side by side: the wave-local kernel (k_spark, qd_chain.h); runtime width, tiles of 1024 samples
template <int F, int TS>
chain_fn pick_spark_ts(int nco) {
    constexpr int NCH = TS / (int)SparkTraits<F>::CH;
    switch (nco) {                              // chains with a shift: register budget of three (two) waves per SIMD, see spark_lb
    case 0: return k_spark<F, 0, DynGeo, NCH, 4>;
    case 1: return k_spark<F, 1, DynGeo, NCH, (TS == 512 ? 3 : 2)>;
    default: return k_spark<F, 2, DynGeo, NCH, 2>;
    }
}
// waves per SIMD the built-in kernel is register-budgeted for (= workgroups per CU): without a shift 128 VGPRs hold everything;
// with one, the lane constants of the row quarters (16 doubles), the NCO's f64 temporaries and the next tile's prefetch beside the
// sixteen-point butterflies need ~150 (tiles of 512 samples) / ~190 (1024): budgeted at 4 they spill 9-85 registers to scratch
int spark_lb(uint32_t ts, int nco) { return nco == 0 ? 4 : (nco == 1 && ts == 512 ? 3 : 2); }
// tile sizes of the built-in (runtime-width) kernels: 1024 samples per wave, which fills the lanes of the sixteen-point base
// butterflies too; chains WITH a shift take 512 up to W = 512 — their lane constants (16 doubles) and the NCO's f64 temporaries on top
// of 1024 samples of prefetch do not fit 128 registers (61-85 spilled), with 512 they do
uint32_t spark_tile(uint32_t W, int nco) { return (nco != 0 && W <= 512) ? 512u : 1024u; }
chain_fn pick_spark(int fmt, int nco, uint32_t ts) {
#define QD_X(F) (ts == 512 ? pick_spark_ts<F, 512>(nco) : pick_spark_ts<F, 1024>(nco))
    switch (fmt) { QD_FMT_CASES(QD_X) }
#undef QD_X
    return nullptr;
}

// ---- shape-specialised kernels (FixedGeo): the chain shapes of BASELINE.json / the README.
// Same source as the generic kernel with W,S,D,T,G as compile-time constants.
const FixedEntry kFixed[] = {
    // configs[1]  "shift 280000 | lowpass -power 20 -decimate 16 2000000 | sparkfft -width 128"   (README.md:57-63)
    // round 3: row-aligned phase 1 (bit 3) WITH non-temporal stream loads (bit 8): 0.232 -> 0.212 ms on one box, 0.233 -> 0.227 on another
    // (profiles/r03/sweep_cfg2_nt.log; round 2 measured the row-aligned phase 1 alone 1.5 % slower, and it still is without nt)
    // + bit 16: the tile's first and last row — the ones the neighbouring tile reads as well — keep the default cache policy (L2 hits
    //   for the second reader), the seven rows in between go non-temporal: 0.200 -> 0.191 ms (profiles/r03/sweep_nt_inner.log)
    QD_FIXED_FB(0, 1, 128, 128, 16, 40, 2, 9, true, 4, 1, 1, 65800, "cfg2"),
    QD_FIXED_FB(0, 2, 128, 128, 16, 40, 2, 9, true, 4, 1, 1, 65800, "cfg2"),
    // north_star target sentence: 200-tap FIR decimate 32 -> 128-pt FFT
    // packed lane-per-output FIR on a 16-byte-row tile (FixedGeo FLAGS_ bit 2, PAD 2): half the VALU instructions of the FIR
    // + row-aligned fast phase 1 (bit 3): buffer loads with a per-tile descriptor, compile-time row offsets
    // + deferred FFT (bit 6, two FFT slots): the previous tile's FFT + epilogue on a wave the FIR leaves idle
    // + nt stream loads (bit 8): the slab is read once; the non-temporal policy measured 1.0-1.2 % over the default on three boxes
    //   (profiles/r03/sweep_cfg3p_load_policy.log; sc0 / sc1 on top of it: nothing)
    // + bit 16: first and last row of the tile (shared with the neighbouring tiles) at the default policy: another 1.0 %
    QD_FIXED_FB(0, 1, 128, 128, 32, 200, 1, 9, true, 4, 2, 2, 65868, "cfg3p"),
    QD_FIXED_FB(0, 2, 128, 128, 32, 200, 1, 9, true, 4, 2, 2, 65868, "cfg3p"),
    // README.md:90-94 / configs[2] / configs[4] (64-pt windows, stride 16, 400 taps): the STREAMING three-stage kernel
    // (k_chain_pipe3s, FLAGS 32 | 256 | 32768 | 131072): eight producer waves, the shared-FIR waves and four FFT waves work one step
    // apart on a contiguous run of tiles per workgroup; shifted samples and decimated outputs are carried from tile to tile in LDS
    // rings, so a step shifts and filters only what is NEW (a 12-window tile of the plain three-stage kernel repeated 31 % of its
    // phase 1 and 25 % of its FIR).  cf32 (cfg5, 16 GiB per GPU): one-tile-per-CU kernel 7.50 ms -> three-stage 6.71 -> streaming,
    // 14-window steps 5.7 ms; cs8 (cfg3): 25.6 -> 22 ms (12-window steps: 16 would need 174 KiB of LDS).  Identical bytes
    // (profiles/r03/sweep_stream.log).  nt = 512: rows of 512 producer threads; the launch adds 512 consumer threads.
    { 0, 1, 64, 16, 32, 400, 14, 4, 512, 2, 1, 164128, 7, true, 8, 1,
      qd::k_chain_pipe3s<0, 1, qd::FixedGeo<64, 16, 32, 400, 14, 8, 1, 2, 1, 164128>, 7, 4>, "fsk5" },
    { 0, 2, 64, 16, 32, 400, 14, 4, 512, 2, 1, 164128, 7, true, 8, 1,
      qd::k_chain_pipe3s<0, 2, qd::FixedGeo<64, 16, 32, 400, 14, 8, 1, 2, 1, 164128>, 7, 4>, "fsk5" },
#ifndef QD_DEV_FAST
    // cs8 input (HackRF) of the same chain: four producer waves (rows of 256 threads x 4 samples = 1024 samples, seven per step) so that
    // 14-window steps start on row boundaries like the cf32 form's (rows of 2048 samples admit 12 or 16 windows, and 16 do not fit)
    { 1, 1, 64, 16, 32, 400, 14, 4, 256, 2, 1, 164128, 7, true, 8, 1,
      qd::k_chain_pipe3s<1, 1, qd::FixedGeo<64, 16, 32, 400, 14, 8, 1, 2, 1, 164128>, 7, 4, 256>, "cfg3" },
    { 1, 2, 64, 16, 32, 400, 14, 4, 256, 2, 1, 164128, 7, true, 8, 1,
      qd::k_chain_pipe3s<1, 2, qd::FixedGeo<64, 16, 32, 400, 14, 8, 1, 2, 1, 164128>, 7, 4, 256>, "cfg3" },
#endif
    // configs[3]  512-tap FIR decimate 8 -> 1024-pt FFT (no shift)
    // 70 KiB tile: one workgroup per CU, so give it 1024 threads (16 waves/CU); 5 rows of 2048 samples
    // FLAGS 128 (kGeoPackedTile): the two-outputs-per-lane FIR as straight-line packed code, truncated outputs as in-chain
    // snapshots (no helper wave): 37.1 -> 24.9 ms
    // + FLAGS 64 with two FFT slots: the previous window's FFT + epilogue on four of the eight waves the FIR leaves idle -> 23.9 ms
    // + FLAGS 8: row-aligned phase 1 (a window is four rows of 2048 samples + 512) -> 23.3 ms
    // round 3, FLAGS 8192: HALF-window tiles — the raw buffer holds the input of 512 outputs, a window is filtered in two passes into
    // one FFT slot; 512 threads (4 FIR waves + the four-wave deferred FFT), 74 KiB, so TWO workgroups share a CU and one's phase 1 +
    // barriers (8 k of its 29 k cycles per window) run under the other's FIR: 23.1 -> 22.0 ms, now at the package power cap too
    // (1390 W; the clock went from 2.32 to 2.19 GHz — profiles/r03/sweep_cfg4_half.log)
    { 0, 0, 1024, 1024, 8, 512, 1, 4, 512, 2, 2, 8392, 5, true, 4, 2,
      qd::k_chain<0, 0, qd::FixedGeo<1024, 1024, 8, 512, 1, 4, 2, 2, 2, 8392>, true, 5, true, true, 4, 512>, "cfg4" },
};

const FixedEntry *find_fixed(int fmt, int nco, uint32_t W, uint32_t S, uint32_t D, uint32_t T) {
    for (const FixedEntry &e : kFixed)
        if (e.fmt == fmt && e.nco == nco && e.W == W && e.S == S && e.D == D && e.T == T) return &e;
    return nullptr;
}

// ---- plan-time specialisation (hiprtc): any chain shape gets a FixedGeo build of the same kernel source.
// The headers are read from <dir of this .so>/csrc (the in-tree layout); compiled modules are cached per
// process.  QD_JIT=0 disables, QD_JIT=1 forces it for every plan; by default only streams whose chain
// input is >= 16 MiB pay the ~0.3 s compile.
struct JitKey {
    int fmt, nco, fir, rch, whole, lb, nt; uint32_t W, S, D, T, G; uint32_t firb = 8, firr = 1; int noslp = 0; uint32_t pad = 1, batch = 1, flags = 0;
    uint64_t taps_hash = 0;        // baked-taps builds: FNV-1a of the filter (the code depends on it)
    int epi = 0;                   // kernels that take the sink as a template argument (k_spark2)
    bool operator<(const JitKey &o) const {
        return std::tie(fmt, nco, fir, rch, whole, lb, nt, W, S, D, T, G, firb, firr, noslp, pad, batch, flags, taps_hash, epi) <
               std::tie(o.fmt, o.nco, o.fir, o.rch, o.whole, o.lb, o.nt, o.W, o.S, o.D, o.T, o.G, o.firb, o.firr, o.noslp, o.pad, o.batch, o.flags, o.taps_hash, o.epi);
    }
};
std::mutex g_jit_mu;
// A hipFunction_t belongs to the module hipModuleLoadData loaded on the device that was current then: one entry per (device,
// key).  The on-disk code object is shared, only the load is repeated per device (one-process multi-device plans,
// qd_plan_options.shard_device[]; that path is unexercised until a multi-GPU box is available — DESIGN.md section 8).
std::map<std::pair<int, JitKey>, hipFunction_t> g_jit_cache;
// builds that FAILED in this process (a static_assert of the kernel the host's restatement of its geometry did not foresee, a hiprtc
// error): remembered with their message, so that every later plan of the shape falls back at once instead of paying the compile again
std::map<std::pair<int, JitKey>, std::string> g_jit_failed;

std::string csrc_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&csrc_dir), &info) || !info.dli_fname) return "";
    std::string path(info.dli_fname);
    size_t slash = path.rfind('/');
    return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/csrc";
}

// ---- on-disk cache of plan-time builds: a compile costs ~0.3-1 s, which a single pass over anything smaller than tens
// of GiB never repays; a cached code object loads in ~1 ms.  Directory: $QD_JIT_CACHE ("0" / "off" disables), else
// $XDG_CACHE_HOME/quadrs_hip, else $HOME/.cache/quadrs_hip.  File name: FNV-1a of (kernel name, options, the two kernel
// headers' contents); file = "QDJIT1\n<lowered name>\n" + code object.  Every failure just means "not cached".
uint64_t fnv1a(const void *data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
bool read_file(const std::string &path, std::vector<char> *out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out->resize((size_t)n);
    const bool ok = n == 0 || fread(out->data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
std::string jit_cache_dir() {
    std::string d;
    if (const char *e = getenv("QD_JIT_CACHE")) {
        if (!*e || !strcmp(e, "0") || !strcmp(e, "off")) return "";
        d = e;
    } else if (const char *x = getenv("XDG_CACHE_HOME")) { if (*x) d = std::string(x) + "/quadrs_hip"; }
    if (d.empty()) { const char *h = getenv("HOME"); if (!h || !*h) return ""; d = std::string(h) + "/.cache/quadrs_hip"; }
    for (size_t i = 1; i <= d.size(); ++i)                       // mkdir -p; errors surface as "cannot write" later
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0755);
    return d;
}

// returns nullptr (and leaves a message in *why) when specialisation is not possible; with may_compile false only the
// in-process and on-disk caches are consulted
hipFunction_t jit_chain_kernel(const JitKey &k, std::string *why, bool may_compile = true, const std::vector<float> *taps = nullptr) {
    std::lock_guard<std::mutex> lock(g_jit_mu);
    int jit_dev = 0;
    (void)hipGetDevice(&jit_dev);
    const std::pair<int, JitKey> dk(jit_dev, k);
    auto it = g_jit_cache.find(dk);
    if (it != g_jit_cache.end()) return it->second;
    if (auto bad = g_jit_failed.find(dk); bad != g_jit_failed.end()) { *why = bad->second; return nullptr; }
    const std::string dir = csrc_dir();
    std::vector<char> hdr1, hdr2;
    if (!read_file(dir + "/qd_chain.h", &hdr1) || !read_file(dir + "/qd_device.h", &hdr2)) {
        *why = "kernel headers not found next to the library (" + dir + ")";
        return nullptr;
    }
    char name[512];
    if ((k.flags & kGeoSpark) && (k.flags & kGeoSparkDirect))   // ... overlapping windows of 2 ... 8 points, a window per lane
        snprintf(name, sizeof name, "qd::k_spark0<%d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d>", k.fmt, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.lb, k.epi);
    else
    if ((k.flags & kGeoSpark) && (k.flags & kGeoSparkReg))      // ... with the first FFT pass out of registers (cf32, W = 128 ... 1024)
        snprintf(name, sizeof name, "qd::k_spark2<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d>", k.fmt, k.nco, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.lb, k.epi);
    else
    if (k.flags & kGeoSpark)       // the wave-local kernel of chains without a lowpass: rch = chunks per tile
        snprintf(name, sizeof name, "qd::k_spark<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d, %d>", k.fmt, k.nco, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.rch, k.lb, k.epi);
    else
    if ((k.flags & kGeoPipe3) && (k.flags & kGeoStream))       // ... its streaming form: contiguous runs of tiles, state carried in LDS rings
        snprintf(name, sizeof name, "qd::k_chain_pipe3s<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d, %d>", k.fmt, k.nco, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.rch, k.lb, k.nt);
    else
    if (k.flags & kGeoPipe3)       // the three-stage kernel for overlapping windows (k_chain_pipe3): 512 producer threads + FIR + FFT waves
        snprintf(name, sizeof name, "qd::k_chain_pipe3<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d>", k.fmt, k.nco, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.rch, k.lb);
    else
    if (k.flags & kGeoPipe)        // the role-split kernel (k_chain_pipe): 256 producer threads + one consumer wave
        snprintf(name, sizeof name, "qd::k_chain_pipe<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %d, %d, %d>", k.fmt, k.nco, k.W,
                 k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.rch, k.lb, (k.flags & kGeoPipeFftWave) ? 384 : 320);
    else
    snprintf(name, sizeof name, "qd::k_chain<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u, %u, %u>, %s, %d, %s, true, %d, %d>", k.fmt, k.nco, k.W,
             k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.batch, k.flags, k.fir ? "true" : "false", k.rch, k.whole ? "true" : "false", k.lb, k.nt);
    std::string src;
    if ((k.flags & kGeoBakedTaps) && taps && !taps->empty()) {       // the plan's filter as exact hex-float literals
        src += "#define QD_BAKED_TAPS_LIST ";
        char lit[48];
        for (size_t i = 0; i < taps->size(); ++i) { snprintf(lit, sizeof lit, "%s%af", i ? ", " : "", (double)(*taps)[i]); src += lit; }
        src += "\n";
    }
    src += std::string("#include \"qd_chain.h\"\ntemplate __global__ void ") + name + "(const qd::ChainParams);\n";
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "qd_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *why = "hiprtcCreateProgram failed"; return nullptr; }
    hiprtcAddNameExpression(prog, name);
    const std::string inc = "-I" + dir;
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", inc.c_str(),
                          "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"};      // see build.py: the tile queue's claim stays one plain atomic
    std::vector<const char *> optv(opts, opts + sizeof opts / sizeof opts[0]);
#ifdef QD_STAMP
    optv.push_back("-DQD_STAMP");
#endif
#ifdef QD_DEVELOP
    optv.push_back("-DQD_DEVELOP");
#endif
#ifdef QD_WGTIME
    optv.push_back("-DQD_WGTIME");
#endif
    if (k.noslp || dev_env("QD_JIT_NOSLP")) optv.push_back("-fno-slp-vectorize");   // scalar f32 accumulate chains of the long-filter policy
    std::vector<std::string> extra;                          // development: QD_JIT_FLAGS="-mllvm -foo ..." appended verbatim
    if (const char *e = dev_env("QD_JIT_FLAGS")) {
        std::string cur;
        for (const char *c = e;; ++c) {
            if (*c == ' ' || *c == 0) { if (!cur.empty()) extra.push_back(cur); cur.clear(); if (!*c) break; }
            else cur.push_back(*c);
        }
        for (const std::string &x : extra) optv.push_back(x.c_str());
    }
    // cache lookup (the -I path is excluded from the key: the headers' contents are in it)
    std::string cache_file;
    {
        uint64_t h = fnv1a(name, strlen(name));
        h = fnv1a(src.data(), src.size(), h);                  // includes the baked filter, if any
        int rtc_major = 0, rtc_minor = 0;                      // a code object does not outlive the compiler that made it
        (void)hiprtcVersion(&rtc_major, &rtc_minor);
        h = fnv1a(&rtc_major, sizeof rtc_major, h);
        h = fnv1a(&rtc_minor, sizeof rtc_minor, h);
        for (const char *o : optv) if (o != inc.c_str()) h = fnv1a(o, strlen(o) + 1, h);
        h = fnv1a(hdr1.data(), hdr1.size(), h);
        h = fnv1a(hdr2.data(), hdr2.size(), h);
        const std::string cdir = dev_env("QD_JIT_DUMP") ? std::string() : jit_cache_dir();
        if (!cdir.empty()) { char fn[64]; snprintf(fn, sizeof fn, "/%016llx.co", (unsigned long long)h); cache_file = cdir + fn; }
    }
    if (!cache_file.empty()) {
        std::vector<char> blob;
        if (read_file(cache_file, &blob) && blob.size() > 8 && !memcmp(blob.data(), "QDJIT1\n", 7)) {
            const char *nm = blob.data() + 7, *end = static_cast<const char *>(memchr(nm, '\n', blob.size() - 7));
            if (end) {
                const std::string lowered(nm, end);
                hipModule_t mod; hipFunction_t fn = nullptr;
                if (hipModuleLoadData(&mod, end + 1) == hipSuccess && hipModuleGetFunction(&fn, mod, lowered.c_str()) == hipSuccess) {
                    hiprtcDestroyProgram(&prog);
                    g_jit_cache[dk] = fn;
                    return fn;
                }
            }
        }
    }
    if (!may_compile) { *why = "not cached and the stream is too small to repay a plan-time build"; hiprtcDestroyProgram(&prog); return nullptr; }
    hiprtcResult r = hiprtcCompileProgram(prog, (int)optv.size(), optv.data());
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
        *why = "hiprtc: " + log.substr(0, 300);
        g_jit_failed[dk] = *why;
        hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    const char *lowered = nullptr;
    hiprtcGetLoweredName(prog, name, &lowered);
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    if (const char *dump = dev_env("QD_JIT_DUMP")) {          // development: keep the code object for llvm-objdump
        if (FILE *f = fopen(dump, "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    }
    hipModule_t mod; hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || !lowered || hipModuleGetFunction(&fn, mod, lowered) != hipSuccess) {
        *why = "hipModuleLoadData / GetFunction failed";
        hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    if (!cache_file.empty()) {                               // publish atomically: write aside, then rename
        char tmpn[32]; snprintf(tmpn, sizeof tmpn, ".tmp%d", (int)getpid());
        const std::string tmp = cache_file + tmpn;
        if (FILE *f = fopen(tmp.c_str(), "wb")) {
            bool ok = fwrite("QDJIT1\n", 1, 7, f) == 7 && fputs(lowered, f) >= 0 && fputc('\n', f) != EOF &&
                      fwrite(code.data(), 1, code.size(), f) == code.size();
            ok = fclose(f) == 0 && ok;
            if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
        }
    }
    hiprtcDestroyProgram(&prog);
    g_jit_cache[dk] = fn;
    return fn;
}

struct Geometry {
    uint32_t G = 1, Dp = 1, lds_raw_elems = 0;
    size_t lds_bytes = 0;        // dynamic LDS that fits the main kernel AND the generic kernels (the unaligned-tail launch)
    size_t lds_main = 0;         // the main kernel's own need when it is smaller (half-window tiles): its launch size, and what bounds workgroups per CU
};

// Dynamic LDS of a chain kernel with this tiling.  Layout (qd_chain.h, k_chain prologue): raw tile | batch x G*W FFT buffers |
// twiddles | taps | 8-bit LUT | shared-FIR dec/trc | batch bookkeeping.  The generic kernels (interleaved tile, pad 1, batch 1,
// taps in LDS) run inside the same allocation for the unaligned slab tail, so the size is the larger of the two layouts.
size_t lds_for(uint32_t G, uint64_t W, uint64_t S, uint64_t D, uint64_t T, uint32_t *raw_elems, uint32_t pad_per_row = 1, uint32_t batch = 1,
               bool lut8 = true, uint32_t flags = 0, size_t *main_only = nullptr, int stream_spl = 2 /* samples per lane and row load (streaming kernel) */,
               int stream_nt = 512 /* its producer threads */) {
    uint64_t tile_raw = (uint64_t)(G - 1) * S * D + W * D + T;
    const uint64_t full_raw = tile_raw;
    if ((flags & kGeoHalfTile) && G == 1 && T > 0 && S >= W) tile_raw = (T - T / 2) + (W / 2 - 1) * D + T;     // FixedGeo::kHalfRaw
    auto interleaved = [&](uint32_t padv) {
        const uint64_t pad = (D % 2 == 0) ? padv * (tile_raw / D + 1) : 0;
        uint64_t elems = tile_raw + pad + 1;
        const uint64_t min_elems = (uint64_t)G * W / 2 + 1;     // bucket epilogue parks G*W f32 norms here
        if (elems < min_elems) elems = min_elems;
        return (elems + 1) & ~1ull;                             // keep fb 16-byte aligned
    };
    auto gen_elems_of = [&](uint64_t raw) {                       // the runtime-geometry kernels' tile: pad 1, always the full tile
        const uint64_t pad = (D % 2 == 0) ? (raw / D + 1) : 0;
        uint64_t elems = raw + pad + 1;
        const uint64_t min_elems = (uint64_t)G * W / 2 + 1;
        if (elems < min_elems) elems = min_elems;
        return (elems + 1) & ~1ull;
    };
    const uint64_t shared_fir = (T && S < W) ? 2 * ((uint64_t)(G - 1) * S + W) * 8 : 0;   // dec[] + trc[] of the shared-FIR mode
    const uint64_t taps_b = ((T + 3) & ~3ull) * 4, lut_b = lut8 ? 256 * 4 : 0;
    uint64_t elems = interleaved(pad_per_row);
    // planar tile + immediates for taps (FixedGeo::kPlanar / kBakedTaps; the kernel falls back to the interleaved layout
    // when its geometry conditions fail, which only needs less)
    const bool planar = (flags & kGeoPlanar) && tile_raw < (1u << 24);
    if (planar) { const uint64_t pe = ct_plane_floats((uint32_t)W, (uint32_t)S, (uint32_t)D, (uint32_t)T, G); if (pe > elems) elems = pe; }
    // *raw_elems is what the runtime-geometry kernels read (ChainParams::lds_raw_elems): THEIR raw tile, whatever the main kernel's
    if (raw_elems) *raw_elems = (uint32_t)gen_elems_of(full_raw);
    const bool baked = planar && (flags & kGeoBakedTaps);
    const uint64_t main_b = elems * 8 + (uint64_t)batch * G * W * 8 + W * 8 + (baked ? 0 : taps_b) + lut_b + shared_fir + (uint64_t)batch * 16 + 16;
    const uint64_t generic_b = gen_elems_of(full_raw) * 8 + (uint64_t)G * W * 8 + W * 8 + taps_b + lut_b + shared_fir + 16 + 16;
    if ((flags & kGeoPipe3) && (flags & kGeoStream)) {
        // k_chain_pipe3s (qd_chain.h, Pipe3S): sample ring + mirror | dec + trc rings of 3 G S | G*W FFT buffers | twiddles | taps
        const uint64_t row = (uint64_t)stream_nt * stream_spl, n_new = (uint64_t)G * S * D, dp = D + ((D % 2 == 0) ? pad_per_row : 0);
        const uint64_t rr = (2 * n_new + T + 2 * D + row - 1) / row, ringd = rr * (row / D);
        const uint64_t c = T - T / 2, mird = ((c % D) + T + D - 1) / D + 1;
        const uint64_t raw_e = ((ringd + mird) * dp + 1) & ~1ull;
        const uint64_t p3 = (flags & kGeoWriteSink) ? raw_e * 8 + taps_b + 64      // the write sink: sample ring | taps
                                                    : raw_e * 8 + (S < W ? 2 : 1) * 3 * (uint64_t)G * S * 8 + 2 * (uint64_t)G * W * 8 + W * 8 + taps_b + 64 + 256;      // trc: overlapping windows only; two transform buffers, on a 256-byte boundary
        if (main_only) *main_only = (size_t)p3;
        return (size_t)(p3 > generic_b ? p3 : generic_b);
    }
    if (flags & kGeoPipe3) {
        // k_chain_pipe3: two raw tiles | dec + trc of two sets | G*W FFT buffers | twiddles | taps | queue hand-over
        const uint64_t qp = (((uint64_t)(G - 1) * S + W) + 1) & ~1ull;
        const uint64_t p3 = 2 * elems * 8 + 4 * qp * 8 + (uint64_t)G * W * 8 + W * 8 + taps_b + 64;
        if (main_only) *main_only = (size_t)p3;
        return (size_t)(p3 > generic_b ? p3 : generic_b);
    }
    if (main_only) *main_only = (size_t)main_b;       // what the MAIN kernel needs (half-window tiles: well under the generic kernels' full tile)
    return (size_t)(main_b > generic_b ? main_b : generic_b);
}

constexpr size_t kLdsMax = 160 * 1024;

}  // namespace

// ------------------------------------------------------------------ plan

// One NCO row table (device): RowBase of the rows [row0, row0 + rows) of `row_len` samples.  A table is only ever
// rewritten (k_rowtab into the same buffer) or regrown after the stream that last read it has drained.
struct RowTab {
    RowBase *d = nullptr;
    uint64_t cap = 0, row0 = 0, rows = 0, off = 0;
    bool used = false;
};
// tables one launch context needs: the main kernel's rows and, for plans whose main kernel is not 256 threads wide,
// rows laid out for the 256-thread per-sample kernel that takes the windows at an unaligned slab end
struct NcoTabs {
    RowTab main, tail;
    std::vector<RowTab> phase;               // interleaved launches with a shift: one row grid per launch (offset phi S)
    unsigned long long *work = nullptr;      // the launch context's tile-queue counters (ChainParams::work), zero between launches
    // One launch context = one set of tile-queue counters + row tables, so launches that use it are ORDERED even when they
    // come on different streams: every launch records `done` behind itself, and a launch arriving on another stream waits
    // for it first (hipStreamWaitEvent, device side).  Two kernels of one context therefore never claim tiles from the same
    // counters at the same time, and `done` transitively covers every earlier reader of the row tables.
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool launched = false;
};

struct qd_plan {
    qd_chain_desc d{};
    qd_plan_options opt{};
    int device = 0;
    bool has_shift = false, has_fir = false;
    uint32_t W = 0, logW = 0, S = 0, D = 1, T = 0;
    uint32_t blk_len = 0, blk_subs = 1;     // QD_EPI_CF32_BLOCKS: read_at block length and sub-windows per block
    uint32_t tile_extra = 0;                // ... and extra raw samples per tile (see ChainParams)
    uint64_t dec_len = 0, n_windows = 0, out_rate = 0;
    double ratio = 0.0;
    std::vector<float> taps_h;
    float *taps_d = nullptr;
    FftLayout fft;
    float2 *tw_d = nullptr;
    Geometry geo;
    chain_fn fn = nullptr, fn_unaligned = nullptr;
    const FixedEntry *fixed = nullptr;
    uint32_t spark_ts = 0;               // ... its tile: samples per wave (512, 1024 or 2048)
    int spark_lb = 4;                    // ... waves per SIMD it is register-budgeted for (= workgroups per CU)
    bool spark_jt_lds = false;           // ... plan-time k_spark with a shift: the lane table sits in LDS (8 KiB more)
    uint32_t phase_unit = 1;             // ... window ranges that start on multiples of it start on a load vector
    bool spark_ov = false;               // ... overlapping windows without a lowpass or a shift on the wave-local kernels (plan-time builds only)
    uint32_t spark_R = 0;                // ... as W / S interleaved launches of side-by-side windows (k_spark; stride divides width); <= 1: one launch (k_spark2)
    bool spark = false;                  // the wave-local kernel of chains without a lowpass (k_spark) is this plan's main kernel
    hipFunction_t jit_fn = nullptr;      // plan-time specialised kernel (hiprtc), replaces fn for aligned launches
    std::string jit_note;
    int wg_per_cu = 1, n_cu = 256, prefetch_mode = 2, nco = 0, nt = kThreads;      // nt: threads that share a row of phase 1 (row = nt * SPL samples)
    int launch_nt = kThreads;            // workgroup size of the main kernel: nt, plus the consumer wave of the role-split kernel
    uint32_t kflags = 0;                 // FixedGeo FLAGS_ of the main kernel
    uint32_t dbg = 0;                    // development builds: ablation bits, read once at plan creation
    // NCO tables: lane tables per plan, row tables per launch context (device path; one per slot of the host ring)
    double2 *jtab_d = nullptr, *jtab256_d = nullptr;
    NcoTabs tabs_dev, tabs_slot[2];
    // take_fft mode (generic kernels): per-window start offsets and an f32 window, both on the device
    const uint64_t *row_offsets_d = nullptr;
    const float *window_d = nullptr;
    // timing
    bool timing = false, ev_made = false, ev_recorded = false;
    hipEvent_t ev0{}, ev1{};
    // host streaming
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
    void *dev_in[2] = {nullptr, nullptr}, *dev_out[2] = {nullptr, nullptr};
    size_t stage_in_bytes = 0, stage_out_bytes = 0, pin_in_bytes = 0, pin_out_bytes = 0;
    hipStream_t streams[2] = {nullptr, nullptr};
    qd_plan_stats stats{};
    // composite plans (a window whose FIR input W*D + T exceeds the 160 KiB LDS tile, stride == width): stage A filters and decimates the
    // stream in read_at blocks of W outputs (the write sink's kernels: per-block truncation == the sink's per-window truncation,
    // src/filter.rs:68-83), stage B transforms the W-point windows of that decimated stream; `cmp_tmp` carries it through HBM
    qd_plan *cmp_a = nullptr, *cmp_b = nullptr;
    void *cmp_tmp = nullptr, *cmp_in = nullptr, *cmp_out = nullptr;
    size_t cmp_tmp_bytes = 0, cmp_in_bytes = 0, cmp_out_bytes = 0;
    hipEvent_t cmp_done = nullptr;
    bool cmp_used = false;
    // sharded plans (options.n_shards > 1): one child plan per shard, created on that shard's device
    std::vector<qd_plan *> shards;
    std::vector<qd_shard_info> shard_info;
    std::mutex mu;
};

namespace {

uint64_t out_bytes_per_window(const qd_plan *p) {
    switch (p->d.epilogue) {
    case QD_EPI_NORMS_F32: return (uint64_t)p->W * 4;
    case QD_EPI_GLYPH_U8: return p->W;
    case QD_EPI_CF32_BLOCKS: return (uint64_t)p->blk_len * 8;
    default: return 1;
    }
}

int ensure_rowtab_for(qd_plan *p, uint32_t ROW, RowTab *t, uint64_t n_lo, uint64_t n_hi, hipStream_t st, uint64_t n_off = 0) {
    const uint64_t r_lo = n_lo / ROW, r_hi = (n_hi + ROW - 1) / ROW + 1;
    if (t->d && t->off == n_off && r_lo >= t->row0 && r_hi <= t->row0 + t->rows) { t->used = true; return QD_OK; }
    // The table is about to be rewritten: whatever read it last must have finished.  launch_chain has already ordered `st` behind
    // the context's previous launch (NcoTabs::done — an event, not the earlier caller's stream handle, which may be destroyed by
    // now), so k_rowtab on `st` runs after every earlier reader.
    const uint64_t rows = r_hi - r_lo;
    if (rows > t->cap) {
        if (t->d) { if (t->used) HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(t->d)); t->d = nullptr; t->cap = 0; }
        const uint64_t cap = rows + rows / 8 + 16;            // chunks of a run differ by a row or two: grow once
        HIPCHK(hipMalloc(&t->d, cap * sizeof(RowBase)));
        t->cap = cap;
    }
    t->row0 = r_lo; t->rows = rows; t->used = true; t->off = n_off;
    const uint32_t blocks = (uint32_t)((rows + 255) / 256);
    hipLaunchKernelGGL(k_rowtab, dim3(blocks), dim3(256), 0, st, p->ratio, ROW, r_lo, rows, n_off, t->d);
    HIPCHK(hipGetLastError());
    return QD_OK;
}

int ensure_work(NcoTabs *tabs) {
    if (tabs->work) return QD_OK;
    HIPCHK(hipMalloc(&tabs->work, 9 * 16 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(tabs->work, 0, 9 * 16 * sizeof(unsigned long long)));       // the kernel leaves them zero again
    return QD_OK;
}

void free_rowtab(RowTab *t) {
    if (t->d) (void)hipFree(t->d);
    *t = RowTab{};
}

// What a chain kernel's parameters take from the PLAN (geometry, constants, device tables); the caller adds the call's slab, window range,
// output and row table.
void plan_params(const qd_plan *p, ChainParams *Pp) {
    ChainParams &P = *Pp;
    P.W = p->W; P.logW = p->logW; P.S = p->S; P.D = p->D; P.T = p->T;
    const uint32_t c = p->T - p->T / 2;
    P.G = p->geo.G; P.Dp = p->geo.Dp;
    P.dmagic = p->D > 1 ? (uint32_t)((1ull << 32) / p->D + 1) : 0;
    P.dshift = is_pow2(p->D) ? ilog2(p->D) : 0xffffffffu;
    P.a0 = c / p->D; P.b0 = c % p->D;
    uint32_t tfast = p->D + p->T / 2;
    P.T_fast = tfast < p->T ? tfast : p->T;
    P.a1 = (c + P.T_fast) / p->D; P.b1 = (c + P.T_fast) % p->D;
    P.base_len = p->fft.base_len; P.log_base = p->fft.log_base; P.layers = p->fft.layers;
    P.epi = (uint32_t)p->d.epilogue;
    P.lds_raw_elems = p->geo.lds_raw_elems;
    P.rmin = p->d.has_range ? p->d.range_min : 0.08f;     // src/fft.rs:22-23
    P.rmax = p->d.has_range ? p->d.range_max : 1.0f;
    P.gstep = (P.rmax - P.rmin) / 7.0f;                   // src/fft.rs:45, f32 like the reference
    P.rgstep = 1.0f / P.gstep;                            // glyph_code's short form (qd_device.h)
    P.root2 = (float)std::sqrt(0.5);
    P.tw16_1 = compute_twiddle(1, 16); P.tw16_2 = compute_twiddle(2, 16); P.tw16_3 = compute_twiddle(3, 16);
    P.ratio = p->ratio;
    P.jtab = p->jtab_d; P.taps = p->taps_d; P.tw = p->tw_d;
    P.row_offsets = p->row_offsets_d; P.window = p->window_d;
    P.blk_len = p->blk_len ? p->blk_len : p->W; P.blk_sub_mask = p->blk_subs - 1;
    P.tile_extra = p->tile_extra;
    P.dbg = p->dbg;                                       // 0 except in development builds (QD_DEBUG_SKIP at plan creation)
}

// Overlapping windows without a lowpass or a shift, stride S dividing the width (qd_plan::spark_R = W / S): launch phi covers the windows
// w = phi (mod R) of the range — side by side in the stream that starts phi * S samples later — and writes every R-th output row.
// (Running the R launches over one 8 ... 128 MiB stretch of the stream after the other, so that the re-reads hit the memory-side cache,
// was measured and is slower at every size: profiles/r04/phase_chunk.log.)
int launch_spark_phases(qd_plan *p, NcoTabs *tabs, const void *src_d, uint64_t src_first, uint64_t src_count, uint64_t first_window,
                        uint64_t n_windows, uint64_t out_window0, void *out_d, hipStream_t st) {
    const int bps = bps_of(p->d.format);
    const uint64_t R = p->spark_R, W = p->W, S = p->S, obw = out_bytes_per_window(p);
    if (tabs->launched) HIPCHK(hipStreamWaitEvent(st, tabs->done, 0));
    ChainParams P{};
    plan_params(p, &P);
    P.S = p->W;                                 // each launch's own geometry: windows side by side
    P.lds_dyn = (uint32_t)p->geo.lds_main;
    P.out_row_stride = (uint32_t)R;
    if (p->timing) {
        if (!p->ev_made) { HIPCHK(hipEventCreate(&p->ev0)); HIPCHK(hipEventCreate(&p->ev1)); p->ev_made = true; }
        HIPCHK(hipEventRecord(p->ev0, st));
    }
    const uint64_t cap = (uint64_t)p->n_cu * p->wg_per_cu, last = first_window + n_windows - 1;
    for (uint64_t phi = 0; phi < R; ++phi) {
        if (last < phi) break;
        const uint64_t k_lo = first_window > phi ? (first_window - phi + R - 1) / R : 0, k_hi = (last - phi) / R;
        if (k_hi < k_lo) continue;
        const uint64_t n_phi = k_hi - k_lo + 1, shift_samples = phi * S;
        if (k_lo * W + shift_samples < src_first || (k_hi * W + shift_samples + W) > src_first + src_count)
            return fail(QD_ERR_INVALID, "src slab [%llu,+%llu) does not cover the samples of windows [%llu,+%llu)", (unsigned long long)src_first,
                        (unsigned long long)src_count, (unsigned long long)first_window, (unsigned long long)n_windows);
        P.src = static_cast<const uint8_t *>(src_d) + shift_samples * bps;      // sample n of this launch is sample n + phi S of the stream
        P.src_first = src_first; P.src_count = src_count - shift_samples;
        if (p->has_shift) {
            // the NCO row grid of THIS launch's stream: rows of 512 samples that start phi S samples into the plan's (the caller has checked
            // that the launch's first window sits on it); the lane table does not depend on where a row starts
            if (tabs->phase.size() < R) tabs->phase.resize(R);
            const int rc = ensure_rowtab_for(p, kSparkRow, &tabs->phase[phi], k_lo * W, (k_hi + 1) * W + (uint64_t)P.G * W, st, shift_samples);
            if (rc) return rc;
            P.rowtab = tabs->phase[phi].d; P.rowtab_row0 = tabs->phase[phi].row0;
        }
        P.first_window = k_lo; P.n_windows = n_phi; P.out_window0 = k_lo;
        P.out = static_cast<uint8_t *>(out_d) + ((k_lo * R + phi) - out_window0) * obw;
        const uint64_t n_tiles = (n_phi + P.G - 1) / P.G, wgs = (n_tiles + 3) / 4;
        const uint32_t grid = (uint32_t)(wgs < cap ? wgs : cap);
        void *args[] = {&P};
        HIPCHK(hipModuleLaunchKernel(p->jit_fn, grid, 1, 1, (unsigned)p->launch_nt, 1, 1, (unsigned)p->geo.lds_main, st, args, nullptr));
    }
    if (p->timing) { HIPCHK(hipEventRecord(p->ev1, st)); p->ev_recorded = true; }
    if (!tabs->done) HIPCHK(hipEventCreateWithFlags(&tabs->done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(tabs->done, st));
    tabs->last_stream = st; tabs->launched = true;
    return QD_OK;
}

int launch_chain(qd_plan *p, NcoTabs *tabs, const void *src_d, uint64_t src_first, uint64_t src_count, uint64_t first_window,
                 uint64_t n_windows, uint64_t out_window0, void *out_d, hipStream_t st) {
    if (n_windows == 0) return QD_OK;
    const int fmt = p->d.format;
    const int spl = spl_of(fmt), bps = bps_of(fmt);
    uint64_t need0 = first_window * p->S * p->D;
    uint64_t need1 = (first_window + n_windows - 1) * p->S * p->D + (uint64_t)p->W * p->D + p->T;
    if (!p->row_offsets_d && (need0 < src_first || need1 > src_first + src_count))
        return fail(QD_ERR_INVALID, "src slab [%llu,+%llu) does not cover samples [%llu,%llu) needed by windows [%llu,+%llu)",
                    (unsigned long long)src_first, (unsigned long long)src_count, (unsigned long long)need0,
                    (unsigned long long)need1, (unsigned long long)first_window, (unsigned long long)n_windows);
    int rc = QD_OK;
    bool phases_unaligned = false;
    if (p->spark_R > 1) {
        // interleaved launches of side-by-side windows; a slab that does not start on a load vector goes to the per-sample kernel as ever
        // (with a shift every launch's first window must sit on its NCO row grid: first_window a multiple of R * 512 / W)
        const uint64_t row_w = p->W < kSparkRow ? kSparkRow / p->W : 1;
        if (p->jit_fn && !p->row_offsets_d && (reinterpret_cast<uintptr_t>(src_d) % (spl * bps)) == 0 && (src_first % spl) == 0 &&
            (!p->has_shift || first_window % ((uint64_t)p->spark_R * row_w) == 0))
            return launch_spark_phases(p, tabs, src_d, src_first, src_count, first_window, n_windows, out_window0, out_d, st);
        phases_unaligned = true;
    }
    // see NcoTabs.  ALWAYS wait: a stream handle compared with the previous caller's may be a new stream at a recycled address (the old one
    // destroyed with its kernels still running); a wait on an event recorded in the same stream costs nothing
    if (tabs->launched) HIPCHK(hipStreamWaitEvent(st, tabs->done, 0));
    if (p->has_shift) {
        // row-aligned phase 1: rows of a short last tile's missing windows (and a half-window pass's read-ahead) get table entries too
        // (the streaming kernel parks one more step of rows behind a run's last tile)
        const uint64_t extra = ((p->kflags & (kGeoFastP1 | kGeoPipe3 | kGeoSpark)) && (p->jit_fn || p->fixed || p->spark)) ? (uint64_t)p->geo.G * p->S * p->D * ((p->kflags & kGeoStream) ? 2 : 1) + p->T : 0;
        rc = ensure_rowtab_for(p, p->nt * spl_of(fmt), &tabs->main, need0, need1 + extra, st);
        if (rc) return rc;
    }

    ChainParams P{};
    plan_params(p, &P);
    P.src = static_cast<const uint8_t *>(src_d);
    P.src_first = src_first; P.src_count = src_count;
    P.out_window0 = out_window0;
    P.rowtab = tabs->main.d; P.rowtab_row0 = tabs->main.row0;
    P.out = out_d;
#if defined(QD_STAMP) || defined(QD_WGTIME)
    constexpr size_t kStampWords = 256 + 4 * 4096;
    static unsigned long long *stamps_d = nullptr;
    if (!stamps_d) { HIPCHK(hipMalloc(&stamps_d, kStampWords * 8)); }
    HIPCHK(hipMemsetAsync(stamps_d, 0, kStampWords * 8, st));
    P.stamps = stamps_d;
#endif

    // The aligned kernels issue whole-vector loads: the slab must start on a vector boundary and a
    // window whose last vector would straddle the slab end goes to the per-sample kernel instead.
    const int vec_bytes = spl * bps;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(src_d) % vec_bytes) == 0 && (src_first % spl) == 0 &&
                        src_count * (uint64_t)bps >= (uint64_t)vec_bytes;
    uint64_t n_aligned = 0;
    // row-aligned phase 1 with G S D (not S D) a multiple of the row: the launch's first window must sit on a row boundary too
    bool fast_misaligned = (p->kflags & (kGeoFastP1 | kGeoPipe3)) && (p->jit_fn || p->fixed) && ((first_window * p->S * p->D) % ((uint64_t)p->nt * spl)) != 0;
    // the wave-local kernel: tiles start on NCO rows when the chain shifts, on load vectors otherwise; irregular rows (take_fft) never run on it
    if (p->spark) fast_misaligned = phases_unaligned || p->row_offsets_d != nullptr || ((first_window * p->S) % (p->has_shift ? (uint64_t)kSparkRow : (uint64_t)spl)) != 0;
    if (p->spark && (p->kflags & kGeoSparkDirect) && p->jit_fn) fast_misaligned = p->row_offsets_d != nullptr;      // a window per lane: any window start (S BPS is a multiple of 4)
    if (vec_ok && !fast_misaligned) {
        // windows [first_window, first_window + n_aligned): need-end rounded up to a vector fits in the slab
        const uint64_t step = (uint64_t)p->S * p->D, rpw = (uint64_t)p->W * p->D + p->T;
        const uint64_t usable = (src_count / spl) * spl + src_first;     // end of the last whole vector
        n_aligned = n_windows;
        while (n_aligned > 0 && (first_window + n_aligned - 1) * step + rpw > usable) --n_aligned;
        // The fast phase 1 (FixedGeo FLAGS_ bit 3) loads and parks a compile-time number of rows per tile, whatever the tile's
        // window count: a short LAST tile (n_windows not a multiple of G) costs it a few rows nobody reads — the buffer
        // descriptor's range check covers the slab end, the row table is extended below — instead of a second launch of the
        // per-sample kernel for one window (cfg2: 65 535 windows in tiles of two; ~8 us of a 0.21 ms step).
    }
    const bool tail_tables = n_aligned < n_windows && p->nt != kThreads && p->has_shift;
    if (tail_tables) {
        const uint64_t t0 = (first_window + n_aligned) * p->S * p->D;
        rc = ensure_rowtab_for(p, kThreads * spl, &tabs->tail, t0, need1, st);
        if (rc) return rc;
    }
    const uint64_t cap = (uint64_t)p->n_cu * p->wg_per_cu;
    if (p->timing) {
        if (!p->ev_made) { HIPCHK(hipEventCreate(&p->ev0)); HIPCHK(hipEventCreate(&p->ev1)); p->ev_made = true; }
        HIPCHK(hipEventRecord(p->ev0, st));
    }
    for (int part = 0; part < 2; ++part) {
        const uint64_t w_begin = part == 0 ? first_window : first_window + n_aligned;
        const uint64_t w_count = part == 0 ? n_aligned : n_windows - n_aligned;
        if (w_count == 0) continue;
        P.first_window = w_begin; P.n_windows = w_count;
        if (part == 1 && tail_tables) { P.rowtab = tabs->tail.d; P.rowtab_row0 = tabs->tail.row0; P.jtab = p->jtab256_d; }
        const uint64_t n_tiles = (w_count + P.G - 1) / P.G;
        uint32_t grid = (uint32_t)(n_tiles < cap ? n_tiles : cap);
        if (part == 0 && p->spark) { const uint64_t wgs = (n_tiles + 3) / 4; grid = (uint32_t)(wgs < cap ? wgs : cap); }      // a tile per WAVE, four waves per workgroup
        // dynamic tile queue for the main launch (static strided walk for the short unaligned tail and for tiny grids)
        P.work = nullptr;
        if (part == 0 && !p->spark && (grid & 7u) == 0 && n_tiles >= 4ull * grid) { rc = ensure_work(tabs); if (rc) return rc; P.work = tabs->work; }
        P.lds_dyn = (uint32_t)(part == 0 && p->geo.lds_main ? p->geo.lds_main : p->geo.lds_bytes);
        if (part == 0 && p->jit_fn && !p->row_offsets_d) {
            void *args[] = {&P};
            HIPCHK(hipModuleLaunchKernel(p->jit_fn, grid, 1, 1, (unsigned)p->launch_nt, 1, 1, (unsigned)(p->geo.lds_main ? p->geo.lds_main : p->geo.lds_bytes), st, args, nullptr));
        } else {
            hipLaunchKernelGGL(part == 0 ? p->fn : p->fn_unaligned, dim3(grid), dim3(part == 0 ? p->launch_nt : kThreads), part == 0 && p->geo.lds_main ? p->geo.lds_main : p->geo.lds_bytes, st, P);
            HIPCHK(hipGetLastError());
        }
    }
    if (p->timing) { HIPCHK(hipEventRecord(p->ev1, st)); p->ev_recorded = true; }
    if (!tabs->done) HIPCHK(hipEventCreateWithFlags(&tabs->done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(tabs->done, st));
    tabs->last_stream = st; tabs->launched = true;
#ifdef QD_WGTIME
    {
        std::vector<unsigned long long> h(kStampWords);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(h.data(), P.stamps, kStampWords * 8, hipMemcpyDeviceToHost));
        std::vector<double> dur, endt; unsigned long long t_first = ~0ull, t_last = 0; double per_xcc[16] = {0}; int n_xcc[16] = {0};
        for (int b = 0; b < 4096; ++b) { const unsigned long long *w = &h[256 + 4 * b]; if (!w[3]) continue; t_first = std::min(t_first, w[0]); t_last = std::max(t_last, w[1]); }
        for (int b = 0; b < 4096; ++b) {
            const unsigned long long *w = &h[256 + 4 * b]; if (!w[3]) continue;
            dur.push_back((w[1] - w[0]) * 0.01); endt.push_back((w[1] - t_first) * 0.01); per_xcc[w[2] & 15] += (w[1] - t_first) * 0.01; n_xcc[w[2] & 15]++;
        }
        std::sort(dur.begin(), dur.end()); std::sort(endt.begin(), endt.end());
        if (!dur.empty()) {
            fprintf(stderr, "[wgtime] %zu workgroups, kernel span %.1f us; workgroup run time us: min %.1f median %.1f max %.1f; finish time us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f\n",
                    dur.size(), (t_last - t_first) * 0.01, dur.front(), dur[dur.size() / 2], dur.back(), endt.front(), endt[endt.size() / 10], endt[endt.size() / 2], endt[endt.size() * 9 / 10], endt.back());
            fprintf(stderr, "[wgtime] mean finish time per XCD:");
            for (int x = 0; x < 16; ++x) if (n_xcc[x]) fprintf(stderr, " xcc%d(%d wgs)=%.1f", x, n_xcc[x], per_xcc[x] / n_xcc[x]);
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef QD_STAMP
    {
        unsigned long long h[160];
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(h, P.stamps, sizeof h, hipMemcpyDeviceToHost));
        static const char *names[8] = {"phase1", "bar1", "fir", "bar2", "fft", "bar3", "epilogue", "bar4"};
        double tiles = (double)h[128];
        unsigned grid_wgs = (unsigned)(((n_windows + P.G - 1) / P.G) < cap ? ((n_windows + P.G - 1) / P.G) : cap);
        fprintf(stderr, "[stamps] tiles/launch %.0f (wgs %u): cycles per tile per wave:", tiles, grid_wgs);
        for (int w = 0; w < 16; ++w) {
            if (w >= 4 && h[w * 8] == 0) continue;                 // workgroups with fewer waves
            fprintf(stderr, "\n   wave%d:", w);
            double tot = 0;
            for (int k = 0; k < 8; ++k) { fprintf(stderr, " %s=%.0f", names[k], h[w * 8 + k] / tiles); tot += h[w * 8 + k] / tiles; }
            fprintf(stderr, "  total=%.0f", tot);
        }
        fprintf(stderr, "\n   wave1 phase-1 split per tile: wait_data=%.0f issue_prefetch=%.0f wait_rowbase=%.0f process=%.0f\n", h[129] / tiles, h[130] / tiles, h[131] / tiles, h[132] / tiles);
        fprintf(stderr, "   wave0 phase-1 split per tile: wait_data=%.0f issue_prefetch=%.0f wait_rowbase=%.0f process=%.0f\n", h[133] / tiles, h[134] / tiles, h[135] / tiles, h[136] / tiles);
    }
#endif
    return QD_OK;
}

void free_streaming(qd_plan *p) {
    for (int i = 0; i < 2; ++i) {
        if (p->pin_in[i]) (void)hipHostFree(p->pin_in[i]);
        if (p->pin_out[i]) (void)hipHostFree(p->pin_out[i]);
        if (p->dev_in[i]) (void)hipFree(p->dev_in[i]);
        if (p->dev_out[i]) (void)hipFree(p->dev_out[i]);
        if (p->streams[i]) (void)hipStreamDestroy(p->streams[i]);
        p->pin_in[i] = p->pin_out[i] = p->dev_in[i] = p->dev_out[i] = nullptr;
        p->streams[i] = nullptr;
    }
    p->stage_in_bytes = p->stage_out_bytes = p->pin_in_bytes = p->pin_out_bytes = 0;
}

}  // namespace

extern "C" {

const char *qd_last_error(void) { return g_err.c_str(); }
const char *qd_version(void) { return "quadrs-hip 0.1 (gfx950)"; }

int qd_device_count(int *count) {
    if (!count) return fail(QD_ERR_INVALID, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return QD_OK;
}

int qd_set_device(int device) {
    HIPCHK(hipSetDevice(device));
    return QD_OK;
}

uint64_t qd_pair_bytes(int fmt) {
    switch (fmt) {
    case QD_FMT_CF32: return 8;
    case QD_FMT_CS8: case QD_FMT_CU8: return 2;
    case QD_FMT_CS16: return 4;
    }
    return 0;
}

double qd_shift_ratio(int64_t frequency, uint64_t sample_rate) {
    return (kPi64 * 2.0) * (double)frequency / (double)sample_rate;   // src/shift.rs:28, src/lib.rs:23
}

int qd_lowpass_design(uint64_t frequency, uint64_t sample_rate, size_t size, float *taps) {
    if (!taps) return fail(QD_ERR_INVALID, "taps is NULL");
    design_taps(frequency, sample_rate, size, taps);
    return QD_OK;
}

int qd_plan_destroy(qd_plan *p);
constexpr int kNeedComposite = 1000;       // plan_init -> qd_plan_create_ex: build the two-stage plan (never leaves the library)

static int plan_init(qd_plan *p, const qd_chain_desc &d, uint64_t len, uint64_t rate) {
    (void)hipGetDevice(&p->device);
    p->has_shift = d.has_shift != 0;
    p->has_fir = d.has_lowpass != 0;
    p->W = (uint32_t)d.width; p->logW = ilog2(d.width); p->S = (uint32_t)d.stride;
    if (d.epilogue == QD_EPI_CF32_BLOCKS) {          // tiles are sub-blocks of <= 256 outputs of a read_at block of d.width
        p->blk_len = (uint32_t)d.width;
        p->W = p->blk_len < 256 ? p->blk_len : 256;
        const uint32_t c = (uint32_t)(d.taps - d.taps / 2);
        p->tile_extra = c > d.decimate ? c - (uint32_t)d.decimate : 0;
        // a sub-block's FIR input (W D + T + extra samples) must fit the LDS tile: long decimations take shorter sub-blocks
        while (p->W > 1 && lds_for(1, p->W, p->W, (uint64_t)d.decimate, d.taps + p->tile_extra, nullptr, 1, 1, d.format == QD_FMT_CS8 || d.format == QD_FMT_CU8) > kLdsMax) p->W /= 2;
        p->logW = ilog2(p->W); p->S = p->W;
        p->blk_subs = p->blk_len / p->W;
    }
    p->D = p->has_fir ? (uint32_t)d.decimate : 1;
    p->T = p->has_fir ? (uint32_t)d.taps : 0;
    p->dec_len = len; p->out_rate = rate;
    uint64_t lim = len >= d.width ? len - d.width : 0;
    if (d.epilogue == QD_EPI_CF32_BLOCKS) p->n_windows = (d.n_samples - d.taps) / (d.width * d.decimate);   // full read_at blocks
    else if (d.epilogue == QD_EPI_BUCKET2_U8) p->n_windows = lim / d.stride;            // src/fft.rs:86
    else p->n_windows = lim == 0 ? 0 : (lim - 1) / d.stride + 1;                        // src/fft.rs:28,65
    p->ratio = p->has_shift ? qd_shift_ratio(d.shift_hz, d.sample_rate) : 0.0;

    // |place| = n*|ratio| over the whole stream decides the NCO order once per plan: the dropped
    // second-order term is r^2/2 with |r| <= ulp(place)/2; below 2^28 rad that is <= 1.1e-16, inside the
    // scheme's ~4e-16 error budget, above it the second-order kernel is used
    p->nco = !p->has_shift ? 0 : ((std::fabs(p->ratio) * (double)d.n_samples > 268435456.0) ? 2 : 1);
    if (p->has_shift && (p->opt.nco_order == 1 || p->opt.nco_order == 2)) p->nco = p->opt.nco_order;
    if (const char *e = dev_env("QD_DEBUG_SKIP")) p->dbg = (uint32_t)atoi(e);     // development builds: timing-only ablation
    const int policy = p->opt.kernel_policy;

    // tile geometry: a shape-specialised kernel dictates G; otherwise pick G for LDS / lane use
    uint32_t G = 1, raw_elems = 0;
    const uint32_t T_lds = p->T + p->tile_extra;     // LDS sizing sees the extended tile
    if (lds_for(1, p->W, p->S, p->D, T_lds, &raw_elems, 1, 1, d.format == QD_FMT_CS8 || d.format == QD_FMT_CU8) > kLdsMax) {
        // LowPass::read_at allocates whatever buf.len() * D + T asks for (src/filter.rs:68-69); one workgroup's LDS does not.  Windows that
        // lie side by side run as a two-stage (composite) plan instead — qd_plan_create_ex builds it on this status
        if (p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && p->S == p->W) return kNeedComposite;
        return fail(QD_ERR_UNSUPPORTED, "one window (W*D+T = %llu samples) exceeds the 160 KiB LDS tile and the windows overlap or leave gaps (stride != width)",
                    (unsigned long long)((uint64_t)d.width * (d.has_lowpass ? d.decimate : 1) + (d.has_lowpass ? d.taps : 0)));
    }
    p->fixed = (p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && policy != QD_KERNEL_GENERIC) ? find_fixed(d.format, p->nco, p->W, p->S, p->D, p->T) : nullptr;
    // qd_plan_options.tile_hint = {G, NT, FIRR, FIRB, LB, PAD}: force a plan-time build with this tiling instead of the table /
    // heuristics (LB = waves per SIMD the build is register-budgeted for: 4 -> 128 VGPRs, 2 -> 256; PAD = LDS pad elements per row)
    // [6] = tiles per FFT batch (FixedGeo::kBatch), [7] = workgroups per CU (0: as many as LDS / registers admit, at most 4)
    const bool lut8 = d.format == QD_FMT_CS8 || d.format == QD_FMT_CU8;
    // chains without a lowpass whose windows lie side by side: the wave-local kernel (k_spark), for every width it holds in a tile
    p->spark = !p->has_fir && p->S == p->W && p->W <= kSparkMaxW && d.epilogue != QD_EPI_CF32_BLOCKS && policy != QD_KERNEL_GENERIC;
    // ... and OVERLAPPING windows without a lowpass or a shift (`sparkfft -width 4 -stride 2`: README example 1, BASELINE configs[0]'s chain),
    // as plan-time builds only:
    //   * W = 128 ... 1024, any stride: ONE launch of k_spark2 built for the stride — every window's rows are loaded for it, the overlap
    //     comes out of the caches;
    //   * W = 2 ... 8, any stride: k_spark0 — a lane owns a window from load to store (one base butterfly), no LDS;
    //   * the widths between whose stride divides them: the windows phi, phi + R, phi + 2R ... (R = W / S) lie side by side in the stream
    //     shifted by phi * S samples, so the chain is R launches of k_spark, each writing every R-th output row (the row stride lives in
    //     its lean epilogue: norms and glyph sinks).
    // Everything else stays on k_chain.
    if (!p->has_fir && p->S < p->W && p->W <= kSparkMaxW && d.epilogue != QD_EPI_CF32_BLOCKS &&
        policy != QD_KERNEL_GENERIC && policy != QD_KERNEL_NO_PLAN_TIME && ((uint64_t)p->S * bps_of(d.format)) % 4 == 0) {
        // k_spark0: a window per lane, no LDS.  (W = 16 builds and is bit-exact too, but a lane's 64-byte output piece makes quarter-filled
        // store instructions: 2^28 cf32 samples at S = 4 took 5.2 ms against 2.6 for the interleaved launches)
        const bool direct = !p->has_shift && p->W >= 2 && p->W <= 8 && ((uint64_t)p->W * bps_of(d.format)) % 4 == 0;
        const bool one_launch = !p->has_shift && (p->W == 128 || p->W == 256 || p->W == 512 || p->W == 1024);
        const bool phases = p->W % p->S == 0 && p->W / p->S <= 32 && (d.epilogue == QD_EPI_NORMS_F32 || d.epilogue == QD_EPI_GLYPH_U8) && p->W >= (uint32_t)spl_of(d.format);
        if (direct || one_launch || phases) { p->spark = true; p->spark_ov = true; p->spark_R = phases ? p->W / p->S : 1; }
    }
    uint32_t tune[8] = {0, 0, 1, 8, 4, 1, 1, 0};
    uint32_t hint_flags = 0;
    bool tuned = false;
    if (p->opt.tile_hint[0] || p->opt.tile_hint[1]) {
        const uint32_t *h = p->opt.tile_hint;
        const uint32_t t8[8] = {h[0], h[1], h[2] ? h[2] : 1u, h[3] ? h[3] : 8u, h[4] ? h[4] : 4u, h[5] ? h[5] : 1u, (h[6] & 0xffu) ? (h[6] & 0xffu) : 1u, h[7]};
        hint_flags = h[6] >> 8;          // bits 8+ of slot 6: kernel variant flags (1 planar LDS tile, 2 taps baked into the code)
        if (!(p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && t8[4] >= 1 && t8[4] <= 8 && t8[0] >= 1 &&
              (t8[1] == 256 || t8[1] == 512 || t8[1] == 1024) && (t8[5] == 1 || t8[5] == 2) && t8[6] <= 64 && t8[7] <= 8 && hint_flags <= 524287 &&
              lds_for(t8[0], p->W, p->S, p->D, T_lds, nullptr, t8[5], t8[6], lut8, hint_flags, nullptr, spl_of(d.format), (int)t8[1]) <= kLdsMax))
            return fail(QD_ERR_INVALID, "tile_hint {%u,%u,%u,%u,%u,%u,%u,%u} does not fit this chain", t8[0], t8[1], t8[2], t8[3], t8[4], t8[5], t8[6], t8[7]);
        for (int i = 0; i < 8; ++i) tune[i] = t8[i];
        tuned = true;
        p->fixed = nullptr;
    }
    // Plan-time specialisation is wanted for shapes without a built-in kernel once the stream is big enough to
    // repay the ~0.3 s compile (qd_plan_options.kernel_policy: QD_KERNEL_SPECIALISE always, QD_KERNEL_NO_PLAN_TIME /
    // QD_KERNEL_GENERIC never).
    const uint64_t in_bytes = (uint64_t)d.n_samples * bps_of(d.format);
    // (the write sink, QD_EPI_CF32_BLOCKS, has exactly one specialised kernel: the streaming one, chosen further down from the geometry)
    const bool write_sink = d.epilogue == QD_EPI_CF32_BLOCKS;
    const bool jit_ok = policy != QD_KERNEL_GENERIC && policy != QD_KERNEL_NO_PLAN_TIME;
    // a cached build is always used; a NEW build only when forced or when the stream is at least 1 GiB
    const bool may_compile = policy == QD_KERNEL_SPECIALISE || tuned || in_bytes >= (1ull << 30) || d.mode == QD_MODE_FAST;
    uint32_t batch = 1, kflags = 0;    // tiles per FFT batch / variant flags the main kernel is built with (FixedGeo BATCH_, FLAGS_)
    if (p->has_fir) { p->taps_h.resize(p->T); design_taps(d.lowpass_hz, d.sample_rate, p->T, p->taps_h.data()); }
    auto make_key = [&](uint32_t g, int nt, int lb, int noslp, uint32_t padv, uint32_t batchv = 1, uint32_t flagsv = 0) {
        const uint64_t ROW = (uint64_t)nt * spl_of(d.format);
        uint64_t tile_raw = (uint64_t)(g - 1) * p->S * p->D + (uint64_t)p->W * p->D + p->T;
        if ((flagsv & kGeoHalfTile) && g == 1 && p->S >= p->W) tile_raw = (p->T - p->T / 2) + (uint64_t)(p->W / 2 - 1) * p->D + p->T;      // rows of ONE pass
        // a run may start at any window, so a tile starts on a row boundary only if S*D is a multiple of ROW
        const bool tiles_on_rows = (((uint64_t)p->S * p->D) % ROW) == 0 || ((flagsv & (kGeoFastP1 | kGeoPipe3)) && (((uint64_t)g * p->S * p->D) % ROW) == 0);
        uint64_t rows = (tile_raw + ROW - 1) / ROW + (tiles_on_rows ? 0 : 1);
        if ((flagsv & kGeoPipe3) && (flagsv & kGeoStream)) rows = ((uint64_t)g * p->S * p->D) / ROW;      // rows per step of the streaming kernel
        if ((flagsv & kGeoUnrolledFir) && !(flagsv & kGeoPipe3)) noslp = 1;       // its scalar accumulate chains must stay scalar (the three-stage kernel's FIR is the packed asm form)
        return JitKey{d.format, p->nco, p->has_fir ? 1 : 0, rows <= 10 ? (int)rows : 4, rows <= 10 ? 1 : 0, lb, nt,
                      p->W, p->S, p->D, p->T, g, tune[3], tune[2], noslp, padv, batchv, flagsv,
                      (flagsv & kGeoBakedTaps) ? fnv1a(p->taps_h.data(), p->taps_h.size() * sizeof(float)) : 0ull};
    };
    // FIR-dominated shapes (>= 8 taps per input sample): a tile's FIR phase is latency-bound — one wave walks
    // all T taps however few outputs the tile has — so take the largest tile with <= 512 FIR outputs that LDS
    // allows, 512 threads, a 256-VGPR budget and scalar accumulate chains (measured 1.3-2.9x over the small-tile
    // default on six such shapes, scripts/policy_probe.py)
    bool auto_variant = false;     // variant flags derived from the geometry, not from the table or a hint
    // ---- kernel variant from GEOMETRY (plan-time builds of shapes without a built-in kernel).  The variants the built-in
    // cfg3' / cfg4 kernels use are predicates of the shape, not of a table entry: a chain of 192 taps or W = 256 gets the same
    // packed FIR, row-aligned phase 1 with non-temporal loads, deferred FFT and (long windows) two-output register tiling
    // (profiles/r03/shape_sweep.log: 0.45-0.47 of the VALU roof on 160 / 192 / 256 / 384-tap neighbours of cfg3', where the
    // long-filter policy below reached 0.30-0.35).  The predicates restate FixedGeo's own static conditions (qd_chain.h), so the
    // build takes the path asked for; if no variant build is to be had the plan falls back to the plain tiling further down.
    struct { bool valid = false; uint32_t G = 1, batch = 1, flags = 0, firr = 1, firb = 8; int nt = kThreads; } autosel;
    // QD_MODE_FAST (qd_chain_desc.mode): the built-in kernels are exact-order; a fused build is the geometry recipe + bit 14, made
    // at plan time whatever the stream's size.  No recipe for the shape (overlapping windows, short filters): the exact kernels.
    const bool fast_mode = d.mode == QD_MODE_FAST && p->has_fir && jit_ok && !tuned && !write_sink;
    const FixedEntry *fixed_exact = p->fixed;
    if (fast_mode) p->fixed = nullptr;
    if (jit_ok && !tuned && !p->fixed && p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && p->S < p->W && (uint64_t)p->T >= 8ull * p->D) {
        // Overlapping windows with a long filter (the long-filter class below): the THREE-STAGE kernel (k_chain_pipe3) where its geometry
        // holds — straight-line shared FIR on 16-byte rows, at most 256 outputs per tile, tiles on row boundaries of 512 producer
        // threads, two tile buffers in LDS — with the largest such tile.  cfg5's shape: 7.50 -> 6.71 ms against the one-tile-per-CU form.
        const uint32_t W = p->W, S = p->S, D = p->D, T = p->T, c_half = T - T / 2;
        const int spl = spl_of(d.format);
        const uint32_t ntrunc = c_half ? (c_half + D - 1) / D - 1 : 0;
        const bool geo_ok = ntrunc <= S && D % 2 == 0 && c_half % 2 == 0 && T % 4 == 0 && (c_half % D) % 2 == 0 && D % 4 == 0 && T >= 32 && D % spl == 0 &&
                            is_pow2(W) && W <= 1024;
        if (geo_ok) {
            const uint64_t ROW = 512ull * spl, step = (uint64_t)S * D;
            uint64_t a = ROW, b = step; while (b) { const uint64_t t = a % b; a = b; b = t; }      // gcd
            const uint32_t g_unit = (uint32_t)(ROW / a);                                              // tiles start on rows when G is a multiple of this
            // (1) the STREAMING form (k_chain_pipe3s): a step adds G S new outputs on G S FIR lanes and G S D new samples; the step with
            // the most outputs that the rings leave room for (the FIR stage is bound by the latency of one wave's pass over the T taps,
            // so outputs per pass is what counts).  Rows of 512 producer threads, or of 256 where the shorter row admits a larger step
            // (the 8-bit formats: four samples per lane).  The conditions restate Pipe3S<>::ok (qd_chain.h).
            uint32_t best_s = 0; int best_nt = 512;
            for (int snt : {512, 256}) {
                const uint64_t SROW = (uint64_t)snt * spl;
                uint64_t a2 = SROW, b2 = step; while (b2) { const uint64_t t = a2 % b2; a2 = b2; b2 = t; }
                const uint32_t su = (uint32_t)(SROW / a2);
                for (uint32_t g = su; g >= 1 && g <= 64 && (uint64_t)g * S <= 256; g += su) {
                    const uint64_t n_new = (uint64_t)g * S * D, gs = (uint64_t)g * S;
                    if (n_new < (uint64_t)c_half + T) continue;
                    const uint64_t f0 = (n_new - c_half - T) / D + 1, mird = ((c_half % D) + T + D - 1) / D + 1;
                    if (!(SROW % D == 0 && f0 <= gs && f0 > W - S && mird * D <= SROW && (uint64_t)(g - 1) * S + W <= 2 * gs && n_new / SROW <= 10)) continue;
                    if (lds_for(g, W, S, D, T_lds, nullptr, 2, 1, lut8, kGeoUnrolledFir | kGeoPipe3 | kGeoStream, nullptr, spl, snt) > kLdsMax) break;
                    if (p->n_windows < g) break;
                    if (g > best_s) { best_s = g; best_nt = snt; }
                }
            }
            // (2) the tile-at-a-time three-stage kernel, where the streaming form's geometry fails
            uint32_t best = 0;
            for (uint32_t g = g_unit; !best_s && g >= 1 && g <= 64 && (uint64_t)(g - 1) * S + W <= 256; g += g_unit) {
                const uint64_t tile_raw = (uint64_t)(g - 1) * S * D + (uint64_t)W * D + T;
                if ((tile_raw + ROW - 1) / ROW > 10) break;
                if (lds_for(g, W, S, D, T_lds, nullptr, 2, 1, lut8, kGeoUnrolledFir | kGeoPipe3) > kLdsMax) break;
                if (p->n_windows < g) break;
                best = g;
            }
            if (best_s) { autosel.valid = true; autosel.G = best_s; autosel.nt = best_nt; autosel.batch = 1; autosel.flags = kGeoUnrolledFir | kGeoPipe3 | kGeoStream | kGeoNtLoads; }
            else if (best) { autosel.valid = true; autosel.G = best; autosel.nt = 512; autosel.batch = 1; autosel.flags = kGeoUnrolledFir | kGeoPipe3; }
        }
    }
    if (jit_ok && !tuned && !p->fixed && p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && p->S >= p->W) {
        const uint32_t W = p->W, S = p->S, D = p->D, T = p->T, c_half = T - T / 2;
        const int spl = spl_of(d.format);
        auto rows_aligned = [&](uint32_t nt, uint32_t g) {            // fast phase 1: tiles start on a row boundary, <= 10 rows, whole-tile prefetch
            const uint64_t ROW = (uint64_t)nt * spl, tile_raw = (uint64_t)(g - 1) * S * D + (uint64_t)W * D + T;
            return ((uint64_t)S * D) % ROW == 0 && (tile_raw + ROW - 1) / ROW <= 10 && D % spl == 0;
        };
        const bool pk_geo = T >= 64 && T % 4 == 0 && D % 4 == 0 && is_pow2(D) && c_half % 2 == 0 && (c_half % D) % 2 == 0;
        const bool tile2_geo = pk_geo && (T / 2) % 4 == 0 && c_half % 4 == 0 && D / 4 <= 8 && T > D + 16 && W % 2 == 0;
        if (tile2_geo && W >= 512 && (uint64_t)T >= 4ull * D) {
            // one long window per tile (cfg4's recipe): two outputs per lane as straight-line packed code with in-chain
            // snapshots, the previous window's FFT + epilogue on idle waves, as many threads as the FIR has lanes for
            // half-window tiles (two passes per window, two workgroups per CU) where a pass is a whole number of rows of 512 threads
            const uint64_t half_raw = (uint64_t)c_half + (uint64_t)(W / 2 - 1) * D + T, ROW512 = 512ull * spl;
            const bool half_ok = W >= 1024 && W % 4 == 0 && ((uint64_t)(W / 2) * D) % ROW512 == 0 && ((uint64_t)S * D) % ROW512 == 0 &&
                                 (half_raw + ROW512 - 1) / ROW512 <= 10 && D % spl == 0;
            const int nt = half_ok ? 512 : (W >= 1024 ? 1024 : 512);
            const uint32_t fl = kGeoPackedTile | kGeoDeferFft | (half_ok ? (kGeoHalfTile | kGeoFastP1 | kGeoNtLoads) : (rows_aligned(nt, 1) ? (kGeoFastP1 | kGeoNtLoads | kGeoNtInner) : 0u));
            if (lds_for(1, W, S, D, T_lds, nullptr, 2, 2, lut8, fl) <= kLdsMax) {
                autosel.valid = true; autosel.G = 1; autosel.nt = nt; autosel.batch = 2; autosel.flags = fl; autosel.firr = 2; autosel.firb = 4;
            }
        } else if (pk_geo && W <= 256) {
            // 64..256 outputs per tile of 256 threads (cfg3' recipe): packed lane-per-output FIR on a 16-byte-row tile; where the
            // FIR leaves a wave idle, the previous tile's FFT + epilogue runs there
            const uint32_t g = W >= 128 ? 1u : 128u / W;
            const bool defer = (g * W) % 64 == 0 && g * W + 64 <= 256;
            const uint32_t fl = kGeoNoSplit | (defer ? kGeoDeferFft : 0u) | (rows_aligned(256, g) ? (kGeoFastP1 | kGeoNtLoads | kGeoNtInner) : 0u);
            const uint32_t bt = defer ? 2u : 1u;
            if (p->n_windows >= g && lds_for(g, W, S, D, T_lds, nullptr, 2, bt, lut8, fl) <= kLdsMax / 2) {      // at least two workgroups per CU
                autosel.valid = true; autosel.G = g; autosel.nt = 256; autosel.batch = bt; autosel.flags = fl;
            }
        }
    }
    if (jit_ok && write_sink && !tuned && p->has_fir) {
        // The `write` sink (N1: shift -> lowpass -> decimated cf32 in read_at blocks): the streaming kernel with producers and FIR waves
        // only — a step is one sub-block of W = min(block, 256) outputs on as many FIR lanes, which store their outputs themselves;
        // truncation is relative to the block (ChainParams::blk_len).  The conditions restate Pipe3S<>::ok for side-by-side windows.
        const uint32_t W = p->W, D = p->D, T = p->T, c_half = T - T / 2;
        const int spl = spl_of(d.format);
        const bool pk_geo = T % 4 == 0 && (T / 2) % 4 == 0 && T / 4 > 3 && D % 4 == 0 && (c_half % D) % 2 == 0 && D % spl == 0 && is_pow2(W) && W <= 256 &&
                            p->blk_len % W == 0 && is_pow2(p->blk_subs);
        for (int snt : {512, 256}) {
            if (!pk_geo || autosel.valid) break;
            const uint64_t SROW = (uint64_t)snt * spl, n_new = (uint64_t)W * D;
            if (n_new % SROW != 0 || SROW % D != 0 || n_new < (uint64_t)c_half + T || n_new / SROW > 10) continue;
            const uint64_t f0 = (n_new - c_half - T) / D + 1, mird = ((c_half % D) + T + D - 1) / D + 1;
            if (f0 > W || mird * D > SROW) continue;
            const uint32_t fl = kGeoNoSplit | kGeoNtLoads | kGeoPipe3 | kGeoStream | kGeoWriteSink;
            if (lds_for(1, W, W, D, T, nullptr, 2, 1, lut8, fl, nullptr, spl, snt) > kLdsMax ||        // the kernel's own layout (its T, no tile_extra) ...
                lds_for(1, W, W, D, T_lds, nullptr, 1, 1, lut8) > kLdsMax) continue;                        // ... and the generic kernels' tile for an unaligned tail
            autosel.valid = true; autosel.G = 1; autosel.nt = snt; autosel.batch = 1; autosel.flags = fl;
        }
    }
    if (fast_mode) {
        if (autosel.valid) autosel.flags |= kGeoFastFma;
        else p->fixed = fixed_exact;                                   // nothing to fuse in: the exact built-in kernel, if any
    }
    // the long-filter policy serves overlapping windows (shared FIR) and whatever the packed variants above do not take
    bool heavy = jit_ok && !write_sink && !tuned && !p->fixed && p->has_fir && (uint64_t)p->T >= 8ull * p->D && !autosel.valid;
    int jit_lb = 4, jit_noslp = 0;
    uint32_t pad = 1;              // LDS pad elements per row the main kernel is built with (FixedGeo PAD_)
    if (heavy) {
        auto outs = [&](uint32_t g) { return p->S < p->W ? (uint64_t)(g - 1) * p->S + p->W : (uint64_t)g * p->W; };
        uint32_t gh = 1;           // 16-byte aligned LDS rows (pad 2): ds_read_b128 sample pairs in the tap loop (FixedGeo::kPad)
        while (gh < 64 && outs(gh + 1) <= 512 && lds_for(gh + 1, p->W, p->S, p->D, T_lds, nullptr, 2, 1, lut8) <= kLdsMax) ++gh;
        if (p->n_windows && gh > p->n_windows) gh = (uint32_t)p->n_windows;
        p->jit_fn = jit_chain_kernel(make_key(gh, 512, 2, 1, 2), &p->jit_note, may_compile);
        heavy = p->jit_fn != nullptr;                  // else: the default tiling below, on whatever kernel is available
        if (heavy) { G = gh; pad = 2; p->nt = 512; jit_lb = 2; jit_noslp = 1; }
    }
    if (heavy) {
    } else if (tuned) {
        G = tune[0];
        p->nt = (int)tune[1];
        jit_lb = (int)tune[4];
        pad = tune[5];
        batch = tune[6];
        kflags = hint_flags;
    } else if (p->fixed) {
        G = p->fixed->G;
        p->nt = p->fixed->nt;
        pad = (uint32_t)p->fixed->pad;
        batch = (uint32_t)p->fixed->batch;
        kflags = (uint32_t)p->fixed->flags;
    } else if (p->spark) {
        p->spark_ts = spark_tile(p->W, p->nco);
        G = p->spark_ts / p->W;                // windows per wave tile (the kernel derives the same number from its tile size)
        p->nt = (int)(kSparkRow / spl_of(d.format));      // NCO rows of 512 samples: row table and lane table are laid out for that
        kflags = kGeoSpark;
        p->spark_lb = spark_lb(p->spark_ts, p->nco);
        if (jit_ok && (p->W == 128 || p->W == 256 || p->W == 512 || p->W == 1024)) {
            // width 16 or 64 columns: the plan-time kernel that runs the base butterflies out of the row registers (k_spark2);
            // tile = 64 lanes x 2 columns x base rows.  Cached builds always, a new one for streams of 1 GiB and more.
            // (the kernel's own geometry is windows side by side: S = W also where the plan's windows overlap, see spark_R)
            const uint32_t fbase = (ilog2(p->W) & 1) ? 8u : 16u, ts2 = 128u * fbase, g2 = ts2 / p->W;
            const int lb2 = p->has_shift ? (fbase == 8 ? 3 : 2) : (fbase == 8 ? 4 : 3);
            // (overlapping windows, spark_R: this kernel takes them in ONE launch — every window's rows are loaded for it, the overlap is
            // served by the caches —, so the stride goes into the build and the interleaved launches are off)
            // (... without a shift; with one the NCO rows are laid out for whole tiles of side-by-side windows: the S = W build, interleaved)
            const bool ov_shift = p->spark_ov && p->has_shift;
            JitKey k{d.format, p->nco, 0, 0, 1, lb2, kThreads, p->W, ov_shift ? p->W : p->S, 1, 0, g2, 8, 1, 0, 1, 1, kGeoSpark | kGeoSparkReg, 0ull, d.epilogue};
            if ((!ov_shift || d.epilogue == QD_EPI_NORMS_F32 || d.epilogue == QD_EPI_GLYPH_U8) && (p->jit_fn = jit_chain_kernel(k, &p->jit_note, may_compile)) != nullptr) {
                p->spark_ts = ts2; G = g2; kflags |= kGeoSparkReg; p->spark_lb = lb2;
                if (!ov_shift) p->spark_R = 0;
            }
        }
        if (p->spark_ov && !p->has_shift && !p->jit_fn && jit_ok && p->W >= 2 && p->W <= 8 && ((uint64_t)p->W * bps_of(d.format)) % 4 == 0) {
            JitKey k{d.format, 0, 0, 0, 1, 4, kThreads, p->W, p->S, 1, 0, 64, 8, 1, 0, 1, 1, kGeoSpark | kGeoSparkDirect, 0ull, d.epilogue};
            if (hipFunction_t f = jit_chain_kernel(k, &p->jit_note, may_compile)) {
                p->jit_fn = f; G = 64; kflags |= kGeoSparkDirect; p->spark_lb = 4; p->spark_R = 0;
            }
        }
        if (p->spark_ov && !p->jit_fn) {
            // interleaved launches need a plan-time build (k_spark with the sink as a template argument); none to be had: back to k_chain
            const int lbj = 4;
            JitKey k{d.format, p->nco, 0, (int)(p->spark_ts / (64u * (uint32_t)spl_of(d.format))), 1, lbj, kThreads,
                     p->W, p->W, 1, 0, G, 8, 1, 0, 1, 1, kGeoSpark, 0ull, d.epilogue};
            p->jit_fn = jit_ok && p->spark_R > 1 ? jit_chain_kernel(k, &p->jit_note, may_compile) : nullptr;
            if (p->jit_fn) { p->spark_lb = lbj; p->spark_jt_lds = p->has_shift; }
            else {
                p->spark = false; p->spark_ov = false; p->spark_R = 0; p->spark_ts = 0; kflags = 0; p->nt = kThreads; G = 1;
                while (G < 64 && (uint64_t)G * p->W < 256 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 40 * 1024) G *= 2;
                while (G < 64 && (uint64_t)G * p->W < 1024 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 36 * 1024) G *= 2;
                if (p->n_windows && G > p->n_windows) { while (G > 1 && G / 2 >= p->n_windows) G /= 2; }
            }
        }
    } else {
        while (G < 64 && (uint64_t)G * p->W < 256 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 40 * 1024) G *= 2;
        while (G < 64 && (uint64_t)G * p->W < 1024 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 36 * 1024) G *= 2;
        // chains without a lowpass (every sample is an FFT input), windows of 128 points and more: 2048 samples per tile while four
        // workgroups still share a CU — fewer barriers per sample (16 GiB cf32, profiles/r03/nofir_rate.log: W = 128 7.16 -> 6.74 ms,
        // W = 256 10.50 -> 9.57, cs16 W = 512 10.18 -> 9.07; 4096 samples per tile: 9.04 at W = 128; W = 64 loses with 32 windows: 8.23 -> 9.65)
        if (!p->has_fir && p->S >= p->W && p->W >= 128) while (G < 64 && (uint64_t)G * p->W < 2048 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 40 * 1024) G *= 2;
        if (p->n_windows && G > p->n_windows) { while (G > 1 && G / 2 >= p->n_windows) G /= 2; }
        if (autosel.valid) {
            G = autosel.G; p->nt = autosel.nt; jit_lb = 4; pad = 2; batch = autosel.batch; kflags = autosel.flags; tune[2] = autosel.firr; tune[3] = autosel.firb;
            auto_variant = true;
        }
    }
    p->geo.G = G;
    p->kflags = kflags;
    p->launch_nt = p->nt + ((kflags & kGeoPipe3) ? ((kflags & kGeoWriteSink) ? 256 : 512) : ((kflags & kGeoPipe) ? ((kflags & kGeoPipeFftWave) ? 128 : 64) : 0));
    if (p->spark) p->launch_nt = kThreads;         // four waves per workgroup whatever the row geometry (p->nt = 512 / SPL only lays out the NCO tables)
    p->geo.lds_bytes = lds_for(G, p->W, p->S, p->D, (kflags & kGeoWriteSink) ? p->T : T_lds, &raw_elems, pad, batch, lut8, kflags, &p->geo.lds_main, spl_of(d.format), p->nt);     // the generic kernels (pad 1, batch 1) fit inside the same allocation
    if (kflags & kGeoWriteSink) {      // the streaming write kernel is laid out for T; the generic kernels' tile (an unaligned tail) for T + tile_extra
        uint32_t re_gen = 0;
        const size_t gen_b = lds_for(G, p->W, p->S, p->D, T_lds, &re_gen, 1, 1, lut8);
        if (gen_b > p->geo.lds_bytes) p->geo.lds_bytes = gen_b;
        raw_elems = re_gen;
    }
    if (!(kflags & (kGeoHalfTile | kGeoPipe3))) p->geo.lds_main = p->geo.lds_bytes;
    if (p->spark) p->geo.lds_main = ((size_t)(p->W < 32 ? 32 : p->W) + 4 * (size_t)p->spark_ts) * 8 +     // twiddles | four waves' transform buffers (k_spark)
                                    ((((kflags & kGeoSparkReg) && p->has_shift) || p->spark_jt_lds) ? (size_t)kSparkRow * 16 : 0);   // plan-time builds with a shift: + the NCO lane table
    if (kflags & kGeoSparkDirect) p->geo.lds_main = 16;                 // k_spark0 uses no LDS (a token size: 0 means "the generic layout's")
    p->geo.lds_raw_elems = raw_elems;
    p->geo.Dp = p->D + ((p->D % 2 == 0) ? 1 : 0);
    if ((uint64_t)raw_elems * p->D >= (1ull << 32)) return fail(QD_ERR_UNSUPPORTED, "tile too large");

    p->fn = p->fixed ? p->fixed->fn : (p->spark ? pick_spark(d.format, p->nco, p->spark_ts) : pick_generic(d.format, p->nco, p->has_fir, true));
    p->fn_unaligned = pick_generic(d.format, p->nco, p->has_fir, false);
    if (!p->fn || !p->fn_unaligned) return fail(QD_ERR_UNSUPPORTED, "no kernel built for this format (QD_DEV_FAST build?)");
    {
        // plan-time specialisation for shapes without a built-in FixedGeo kernel
        if (p->spark && jit_ok && !p->jit_fn) {
            // the same kernel with the width as a compile-time constant (butterfly loops unroll, one base butterfly instead of five,
            // index arithmetic folds): cached builds always, a new one for streams of 1 GiB and more.  Falls back to the built-in
            // runtime-width kernel — same tiling, same bytes.
            // (with the width a constant one base butterfly is compiled instead of five, and with a shift the lane constants come out of an LDS
            // copy of the lane table: 82-112 VGPRs, four waves per SIMD at either tile size)
            const int lbj = 4;
            JitKey k{d.format, p->nco, 0, (int)(p->spark_ts / (64u * (uint32_t)spl_of(d.format))), 1, lbj, kThreads,
                     p->W, p->W, 1, 0, G, 8, 1, 0, 1, 1, kGeoSpark, 0ull, d.epilogue};
            p->jit_fn = jit_chain_kernel(k, &p->jit_note, may_compile);
            if (p->jit_fn) {
                p->spark_lb = lbj; p->spark_jt_lds = p->has_shift;
                if (p->spark_jt_lds) p->geo.lds_main += (size_t)kSparkRow * 16;       // (lds_main was sized above for the built-in kernel)
            }
        }
        const bool want = !heavy && !p->spark && (tuned || (!p->fixed && jit_ok && (!write_sink || auto_variant)));
        if (want) {
            p->jit_fn = jit_chain_kernel(make_key(G, p->nt, jit_lb, jit_noslp, pad, batch, kflags), &p->jit_note, may_compile, &p->taps_h);
            if (tuned && !p->jit_fn) return fail(QD_ERR_UNSUPPORTED, "tile_hint build failed: %s", p->jit_note.c_str());
            if (!p->jit_fn && auto_variant && fast_mode && fixed_exact) {
                // QD_MODE_FAST had set the exact built-in kernel aside for a fused build that is not to be had: back to the built-in
                // (exact) kernel with ITS tiling, not down to the generic kernels
                p->fixed = fixed_exact;
                G = p->fixed->G; p->nt = p->fixed->nt; pad = (uint32_t)p->fixed->pad; batch = (uint32_t)p->fixed->batch; kflags = (uint32_t)p->fixed->flags;
                auto_variant = false;
                uint32_t re2 = 0;
                p->geo.G = G; p->kflags = kflags;
                p->launch_nt = p->nt + ((kflags & kGeoPipe3) ? ((kflags & kGeoWriteSink) ? 256 : 512) : ((kflags & kGeoPipe) ? ((kflags & kGeoPipeFftWave) ? 128 : 64) : 0));
                p->geo.lds_bytes = lds_for(G, p->W, p->S, p->D, T_lds, &re2, pad, batch, lut8, kflags, &p->geo.lds_main, spl_of(d.format), p->nt);
                if (!(kflags & (kGeoHalfTile | kGeoPipe3))) p->geo.lds_main = p->geo.lds_bytes;
                p->geo.lds_raw_elems = re2;
                p->fn = p->fixed->fn;
            } else
            if (!p->jit_fn && auto_variant) {
                // no variant build (not cached and too small a stream to compile for, or the build failed): the plain tiling —
                // which the generic kernels run in as well
                G = 1; uint32_t re2 = 0;
                while (G < 64 && (uint64_t)G * p->W < 256 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 40 * 1024) G *= 2;
                while (G < 64 && (uint64_t)G * p->W < 1024 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr, 1, 1, lut8) <= 36 * 1024) G *= 2;
                if (p->n_windows && G > p->n_windows) { while (G > 1 && G / 2 >= p->n_windows) G /= 2; }
                p->nt = kThreads; jit_lb = 4; pad = 1; batch = 1; kflags = 0; tune[2] = 1; tune[3] = 8; auto_variant = false;
                p->geo.G = G; p->kflags = 0; p->launch_nt = kThreads;
                p->geo.lds_bytes = lds_for(G, p->W, p->S, p->D, T_lds, &re2, 1, 1, lut8, 0);
                p->geo.lds_main = p->geo.lds_bytes;
                p->geo.lds_raw_elems = re2;
                if (!write_sink) p->jit_fn = jit_chain_kernel(make_key(G, p->nt, jit_lb, 0, 1, 1, 0), &p->jit_note, may_compile, &p->taps_h);      // (the write sink's only other kernel is the generic one)
            }
        }
        // A built-in straight-line FIR kernel (scalar chains of overlapping-window shapes) re-specialised with the plan's OWN filter
        // baked in (taps as immediates: no LDS reads, no registers for them; cfg3 27.8 -> 24.6 ms): worth a compile only for
        // streams of several GiB, falls back silently.  The packed lane-per-output FIR does not need it: it multiplies by the
        // taps straight out of the register pairs an LDS read delivers (fir_pair), which measures the same as immediates.
        if (!want && !heavy && !tuned && p->fixed && jit_ok && (p->fixed->flags & kGeoUnrolledFir) && !(p->fixed->flags & kGeoPipe3) &&
            (policy == QD_KERNEL_SPECIALISE || in_bytes >= (4ull << 30))) {
            JitKey k = make_key(G, p->nt, p->fixed->lb, 0, pad, batch, kflags | kGeoBakedTaps);
            // the table's prefetch shape and FIR knobs, not the heuristic ones (e.g. the 4-row chunks of the cf32 FSK kernel)
            k.rch = p->fixed->rch; k.whole = p->fixed->whole; k.firb = (uint32_t)p->fixed->firb; k.firr = (uint32_t)p->fixed->firr;
            k.noslp = (p->fixed->flags & kGeoUnrolledFir) ? 1 : 0;
            p->jit_fn = jit_chain_kernel(k, &p->jit_note, true, &p->taps_h);
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) p->n_cu = prop.multiProcessorCount;
    int by_lds = (int)(kLdsMax / (p->geo.lds_main ? p->geo.lds_main : p->geo.lds_bytes));
    p->wg_per_cu = by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
    if (p->fixed) { int by_regs = p->fixed->lb * 256 / p->fixed->nt; if (by_regs < 1) by_regs = 1; if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }
    if (p->spark) { const int by_regs = p->spark_lb; if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }
    if (!p->fixed && !p->jit_fn && !p->spark) { const int by_regs = dyn_lb(p->nco); if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }      // the generic kernels' own budget
    if (p->launch_nt > kThreads) { int by_threads = 2048 / p->launch_nt; if (p->wg_per_cu > by_threads) p->wg_per_cu = by_threads; }
    if (tuned || heavy || auto_variant) { int by_regs = (jit_lb * 4 * 64) / p->launch_nt; if (by_regs < 1) by_regs = 1; if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }
    if (tuned && tune[7] && (int)tune[7] < p->wg_per_cu) p->wg_per_cu = (int)tune[7];
    if (const char *e = dev_env("QD_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) p->wg_per_cu = v; }      // development builds
    // Dynamic-LDS limit: the kernels are process-global objects shared by every plan, so the attribute is set to the
    // hardware maximum (160 KiB), never to one plan's tile — a later plan with a smaller tile must not lower the limit
    // under a live plan with a larger one (tests/test_gpu_robustness.py::test_two_live_plans_with_different_lds).
    if (p->jit_fn) {
        // the tiling (G, threads, LDS layout) was chosen for THIS kernel: the generic kernels cannot run in it, so no silent fallback
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(p->jit_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax); e != hipSuccess)
            return fail(QD_ERR_HIP, "hipFuncSetAttribute(plan-time kernel, max dynamic LDS %zu): %s", kLdsMax, hipGetErrorString(e));
    }
    for (chain_fn f : {p->fn, p->fn_unaligned}) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
        if (e != hipSuccess)
            return fail(QD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS %zu): %s", kLdsMax, hipGetErrorString(e));
    }

    // constant tables
    p->fft = fft_layout(d.width);
    if (!p->fft.tw.empty()) {
        HIPCHK(hipMalloc(&p->tw_d, p->fft.tw.size() * sizeof(float2)));
        HIPCHK(hipMemcpy(p->tw_d, p->fft.tw.data(), p->fft.tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (p->has_fir) {
        HIPCHK(hipMalloc(&p->taps_d, p->T * sizeof(float)));
        HIPCHK(hipMemcpy(p->taps_d, p->taps_h.data(), p->T * sizeof(float), hipMemcpyHostToDevice));
    }
    if (p->has_shift) {
        const uint32_t ROW = p->nt * spl_of(d.format);
        HIPCHK(hipMalloc(&p->jtab_d, ROW * sizeof(double2)));
        hipLaunchKernelGGL(k_jtab, dim3((ROW + 255) / 256), dim3(256), 0, 0, p->ratio, ROW, p->jtab_d);
        HIPCHK(hipGetLastError());
        if (p->nt != kThreads) {
            const uint32_t ROW256 = kThreads * spl_of(d.format);
            HIPCHK(hipMalloc(&p->jtab256_d, ROW256 * sizeof(double2)));
            hipLaunchKernelGGL(k_jtab, dim3((ROW256 + 255) / 256), dim3(256), 0, 0, p->ratio, ROW256, p->jtab256_d);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipDeviceSynchronize());
    }
    return QD_OK;
}


namespace {
struct DeviceGuard {                       // hipSetDevice is per host thread: run a plan on the device it was made on
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Equal, contiguous, tile-aligned window ranges (quadrs_amd/shard.py::partition states the same rule for the
// one-process-per-GPU path): shard g owns the source samples from its first window's start up to the next shard's
// first window's start, and reads `halo` samples beyond that.
void partition_windows(uint64_t n_windows, uint32_t n_shards, uint64_t step, uint64_t rpw, uint32_t tile, std::vector<qd_shard_info> *out) {
    out->assign(n_shards, qd_shard_info{});
    uint64_t per = (n_windows + n_shards - 1) / n_shards;
    per = (per + tile - 1) / tile * tile;
    const uint64_t total_end = n_windows ? (n_windows - 1) * step + rpw : 0;
    for (uint32_t g = 0; g < n_shards; ++g) {
        qd_shard_info &s = (*out)[g];
        s.w0 = std::min<uint64_t>(n_windows, (uint64_t)g * per);
        s.w1 = std::min<uint64_t>(n_windows, (uint64_t)(g + 1) * per);
        uint64_t own_first = s.w0 * step;
        uint64_t own_end = (g + 1 < n_shards && s.w1 < n_windows) ? s.w1 * step : total_end;
        if (s.w1 == s.w0) { own_first = total_end; own_end = total_end; }
        const uint64_t need_end = s.w1 > s.w0 ? (s.w1 - 1) * step + rpw : own_first;
        s.own_first = own_first;
        s.own_count = own_end > own_first ? own_end - own_first : 0;
        s.halo = need_end > own_end ? need_end - own_end : 0;
    }
}
}  // namespace

int qd_plan_create_ex(const qd_chain_desc *desc, const qd_plan_options *options, qd_plan **out) {
    if (!desc || !out) return fail(QD_ERR_INVALID, "desc/plan is NULL");
    if (desc->struct_size != sizeof(qd_chain_desc)) return fail(QD_ERR_INVALID, "qd_chain_desc size mismatch");
    qd_plan_options opt{};
    opt.struct_size = sizeof opt;
    if (options) {
        if (options->struct_size != sizeof(qd_plan_options)) return fail(QD_ERR_INVALID, "qd_plan_options size mismatch");
        opt = *options;
        if (opt.kernel_policy < QD_KERNEL_AUTO || opt.kernel_policy > QD_KERNEL_NO_PLAN_TIME) return fail(QD_ERR_INVALID, "unknown kernel_policy %d", opt.kernel_policy);
        if (opt.nco_order < 0 || opt.nco_order > 2) return fail(QD_ERR_INVALID, "nco_order must be 0, 1 or 2");
        if (opt.copy_threads > 64) return fail(QD_ERR_INVALID, "copy_threads > 64");
        if (opt.chunk_bytes && (opt.chunk_bytes < (1u << 16) || opt.chunk_bytes > (1ull << 34))) return fail(QD_ERR_INVALID, "chunk_bytes outside [64 KiB, 16 GiB]");
        if (opt.n_shards > QD_MAX_SHARDS) return fail(QD_ERR_INVALID, "n_shards > %d", QD_MAX_SHARDS);
    }
    const qd_chain_desc &d = *desc;
    if (d.format < 0 || d.format > 3) return fail(QD_ERR_INVALID, "unknown format %d", d.format);
    if (d.epilogue < 0 || d.epilogue > 3) return fail(QD_ERR_INVALID, "unknown epilogue %d", d.epilogue);
    if (d.mode != QD_MODE_EXACT && d.mode != QD_MODE_FAST) return fail(QD_ERR_INVALID, "unknown mode %d", d.mode);
    if (d.epilogue == QD_EPI_CF32_BLOCKS && !d.has_lowpass) return fail(QD_ERR_INVALID, "QD_EPI_CF32_BLOCKS needs a lowpass in the chain");
    if (!is_pow2(d.width))
        return fail(QD_ERR_PANIC, "Radix4 requires a power-of-two width (rustfft API contract), got %llu", (unsigned long long)d.width);
    if (d.width > (1u << 20)) return fail(QD_ERR_UNSUPPORTED, "width too large");
    if (d.stride == 0) return fail(QD_ERR_INVALID, "stride 0 never terminates in the reference (src/fft.rs:65)");
    if (d.stride > 0xffffffffull) return fail(QD_ERR_UNSUPPORTED, "stride too large");
    uint64_t len = d.n_samples, rate = d.sample_rate;
    if (d.has_shift) {
        // Shift::new asserts, src/shift.rs:20-24
        int64_t af = d.shift_hz < 0 ? -d.shift_hz : d.shift_hz;
        if (!(af < (int64_t)(d.sample_rate / 2)) || d.sample_rate == 0)
            return fail(QD_ERR_PANIC, "frequency must be under half the sample rate (src/shift.rs:20-24)");
    }
    if (d.has_lowpass) {
        if (d.decimate == 0) return fail(QD_ERR_PANIC, "decimate 0 divides by zero (src/filter.rs:47)");
        if (d.taps < 2) return fail(QD_ERR_PANIC, "lowpass size < 2 underflows (src/filter.rs:74)");
        if (d.taps > 65536 || d.decimate > 65536) return fail(QD_ERR_UNSUPPORTED, "taps/decimate too large");
        if (len < d.taps) return fail(QD_ERR_PANIC, "inner.len() < filter.len() (src/filter.rs:46)");
        len = 1 + (len - d.taps) / d.decimate;     // LowPass::len, src/filter.rs:47
        rate = rate / d.decimate;                  // src/filter.rs:51
    }
    if (d.epilogue != QD_EPI_CF32_BLOCKS && len < d.width) return fail(QD_ERR_PANIC, "len %llu < width %llu: u64 underflow at src/fft.rs:28,86",
                                   (unsigned long long)len, (unsigned long long)d.width);
    if (opt.n_shards > 1) {
        int n_dev = 0;
        HIPCHK(hipGetDeviceCount(&n_dev));
        for (uint32_t g = 0; g < opt.n_shards; ++g)
            if (opt.shard_device[g] < 0 || opt.shard_device[g] >= n_dev)
                return fail(QD_ERR_INVALID, "shard %u: device %d does not exist (%d visible)", g, opt.shard_device[g], n_dev);
    }
    qd_plan *p = new qd_plan();
    p->d = d;
    p->opt = opt;
    int rc = plan_init(p, d, len, rate);
    if (rc == kNeedComposite) {
        if (opt.n_shards > 1) rc = fail(QD_ERR_UNSUPPORTED, "a window larger than the LDS tile runs as a two-stage plan, which is not sharded inside one process");
        else {
            qd_plan_options copt = opt;
            copt.n_shards = 0;
            memset(copt.tile_hint, 0, sizeof copt.tile_hint);
            qd_chain_desc a = d;                         // stage A: the same source chain into read_at blocks of W decimated samples
            a.stride = d.width; a.epilogue = QD_EPI_CF32_BLOCKS; a.has_range = 0;
            rc = qd_plan_create_ex(&a, &copt, &p->cmp_a);
            if (rc == QD_OK) {
                qd_chain_desc b{};                       // stage B: W-point windows side by side over the decimated stream
                b.struct_size = sizeof b;
                b.format = QD_FMT_CF32; b.sample_rate = rate ? rate : 1;
                b.n_samples = d.epilogue == QD_EPI_BUCKET2_U8 ? (p->n_windows + 1) * d.width : p->n_windows * d.width + 1;      // exactly n_windows windows (src/fft.rs:28,65 / :86)
                b.width = d.width; b.stride = d.width; b.epilogue = d.epilogue; b.mode = d.mode;
                b.has_range = d.has_range; b.range_min = d.range_min; b.range_max = d.range_max;
                rc = qd_plan_create_ex(&b, &copt, &p->cmp_b);
            }
            if (rc == QD_OK && (p->cmp_a->n_windows < p->n_windows || p->cmp_b->n_windows != p->n_windows))
                rc = fail(QD_ERR_UNSUPPORTED, "two-stage plan: stage window counts disagree (%llu blocks, %llu / %llu windows)", (unsigned long long)p->cmp_a->n_windows,
                          (unsigned long long)p->cmp_b->n_windows, (unsigned long long)p->n_windows);
            p->geo.G = 1;
        }
    }
    if (rc) { qd_plan_destroy(p); return rc; }
    // sharded plans: the parent describes the whole stream; each shard gets a plan of its own on its device
    const uint32_t n_shards = opt.n_shards > 1 ? opt.n_shards : 1;
    const uint64_t step = (uint64_t)(p->blk_len ? p->blk_len : p->S) * p->D, rpw = (uint64_t)(p->blk_len ? p->blk_len : p->W) * p->D + p->T;
    // API windows of the write sink are whole blocks; interleaved launches take any window range, and keep their speed when it starts on a load vector
    uint32_t tile_api = p->blk_len ? 1u : p->geo.G;
    if (p->spark_R > 1) {
        tile_api = (uint32_t)spl_of(d.format); while (tile_api > 1 && ((uint64_t)(tile_api / 2) * p->S) % spl_of(d.format) == 0) tile_api /= 2;
        if (p->has_shift) tile_api = p->spark_R * (p->W < kSparkRow ? kSparkRow / p->W : 1u);      // ... and with a shift on the launches' NCO row grids
        p->phase_unit = tile_api;
    }
    partition_windows(p->n_windows, n_shards, step, rpw, tile_api, &p->shard_info);
    for (uint32_t g = 0; g < n_shards; ++g) p->shard_info[g].device = n_shards > 1 ? opt.shard_device[g] : p->device;
    if (n_shards > 1) {
        qd_plan_options copt = opt;
        copt.n_shards = 0;
        for (uint32_t g = 0; g < n_shards && rc == QD_OK; ++g) {
            DeviceGuard guard(opt.shard_device[g]);
            qd_plan *c = nullptr;
            rc = qd_plan_create_ex(desc, &copt, &c);
            if (rc == QD_OK) p->shards.push_back(c);
        }
        if (rc) { qd_plan_destroy(p); return rc; }
    }
    *out = p;
    return QD_OK;
}

int qd_plan_create(const qd_chain_desc *desc, qd_plan **out) { return qd_plan_create_ex(desc, nullptr, out); }

int qd_plan_destroy(qd_plan *p) {
    if (!p) return QD_OK;
    for (qd_plan *c : p->shards) (void)qd_plan_destroy(c);
    p->shards.clear();
    if (p->cmp_a) (void)qd_plan_destroy(p->cmp_a);
    if (p->cmp_b) (void)qd_plan_destroy(p->cmp_b);
    p->cmp_a = p->cmp_b = nullptr;
    DeviceGuard guard(p->device);
    (void)hipDeviceSynchronize();
    for (void *q : {p->cmp_tmp, p->cmp_in, p->cmp_out}) if (q) (void)hipFree(q);
    if (p->cmp_done) (void)hipEventDestroy(p->cmp_done);
    free_streaming(p);
    if (p->taps_d) (void)hipFree(p->taps_d);
    if (p->tw_d) (void)hipFree(p->tw_d);
    if (p->jtab_d) (void)hipFree(p->jtab_d);
    if (p->jtab256_d) (void)hipFree(p->jtab256_d);
    for (NcoTabs *t : {&p->tabs_dev, &p->tabs_slot[0], &p->tabs_slot[1]}) { free_rowtab(&t->main); free_rowtab(&t->tail); for (RowTab &q : t->phase) free_rowtab(&q); t->phase.clear(); if (t->work) (void)hipFree(t->work); t->work = nullptr; if (t->done) (void)hipEventDestroy(t->done); t->done = nullptr; t->launched = false; }
    if (p->ev_made) { (void)hipEventDestroy(p->ev0); (void)hipEventDestroy(p->ev1); }
    delete p;
    return QD_OK;
}

int qd_plan_get_info(const qd_plan *p, qd_plan_info *info) {
    if (!p || !info) return fail(QD_ERR_INVALID, "plan/info is NULL");
    memset(info, 0, sizeof *info);
    info->n_windows = p->n_windows;
    info->decimated_len = p->dec_len;
    info->out_sample_rate = p->out_rate;
    info->out_bytes_per_window = out_bytes_per_window(p);
    info->raw_per_window = (uint64_t)(p->blk_len ? p->blk_len : p->W) * p->D + p->T;
    info->raw_step = (uint64_t)(p->blk_len ? p->blk_len : p->S) * p->D;
    info->ratio = p->ratio;
    info->tile_windows = p->spark_R > 1 ? p->phase_unit : p->geo.G;
    if (p->cmp_a) {                                   // two-stage plan: the kernel figures are stage A's (the filter)
        qd_plan_info ia;
        const int rc = qd_plan_get_info(p->cmp_a, &ia);
        if (rc) return rc;
        info->threads = ia.threads; info->lds_bytes = ia.lds_bytes; info->kernel_kind = ia.kernel_kind; info->kernel_flags = ia.kernel_flags;
        return QD_OK;
    }
    info->threads = (uint32_t)p->launch_nt;
    info->lds_bytes = (uint32_t)(p->geo.lds_main ? p->geo.lds_main : p->geo.lds_bytes);
    info->kernel_kind = p->jit_fn ? 2u : ((p->fixed || p->spark) ? 1u : 0u);
    info->kernel_flags = (p->jit_fn || p->fixed || p->spark) ? p->kflags : 0u;
    info->_reserved = 0;
    return QD_OK;
}

int qd_plan_kernel_name(const qd_plan *p, char *buf, size_t cap) {
    if (!p || !buf || cap == 0) return fail(QD_ERR_INVALID, "plan/buf is NULL");
    if (p->cmp_a) {
        char a[256], b[256];
        (void)qd_plan_kernel_name(p->cmp_a, a, sizeof a); (void)qd_plan_kernel_name(p->cmp_b, b, sizeof b);
        snprintf(buf, cap, "two stages: %s | %s", a, b);
        return QD_OK;
    }
    const int fmt = p->d.format;
    char geo[160];
    if (p->jit_fn || p->fixed)
        snprintf(geo, sizeof geo, "FixedGeo<%u, %u, %u, %u, %u, ..., %u>", p->W, p->S, p->D, p->T, p->geo.G, p->kflags);
    else snprintf(geo, sizeof geo, "DynGeo");
    const char *kn = (p->kflags & kGeoSparkDirect) && p->jit_fn ? "qd::k_spark0"
                   : (p->kflags & kGeoSparkReg) && p->jit_fn ? "qd::k_spark2"
                   : p->spark ? "qd::k_spark"
                   : ((p->kflags & kGeoPipe3) && (p->kflags & kGeoStream) && (p->jit_fn || p->fixed)) ? "qd::k_chain_pipe3s"
                   : ((p->kflags & kGeoPipe3) && (p->jit_fn || p->fixed)) ? "qd::k_chain_pipe3"
                   : ((p->kflags & kGeoPipe) && p->jit_fn) ? "qd::k_chain_pipe" : "qd::k_chain";
    snprintf(buf, cap, "%s<fmt %d, nco %d, %s>, %d threads, %s", kn, fmt, p->nco, geo, p->launch_nt,
             p->jit_fn ? "plan-time build" : (p->fixed || p->spark ? "built-in" : "generic"));
    return QD_OK;
}

int qd_plan_get_taps(const qd_plan *p, float *taps, size_t cap) {
    if (!p || !taps) return fail(QD_ERR_INVALID, "plan/taps is NULL");
    if (p->cmp_a) return qd_plan_get_taps(p->cmp_a, taps, cap);
    if (cap < p->taps_h.size()) return fail(QD_ERR_INVALID, "taps buffer too small");
    if (!p->taps_h.empty()) memcpy(taps, p->taps_h.data(), p->taps_h.size() * sizeof(float));
    return QD_OK;
}

int qd_plan_src_range(const qd_plan *p, uint64_t first_window, uint64_t n_windows, uint64_t *first, uint64_t *count) {
    if (!p || !first || !count) return fail(QD_ERR_INVALID, "NULL argument");
    const uint64_t step = (uint64_t)(p->blk_len ? p->blk_len : p->S) * p->D;
    const uint64_t rpw = (uint64_t)(p->blk_len ? p->blk_len : p->W) * p->D + p->T;
    *first = first_window * step;
    *count = n_windows ? (n_windows - 1) * step + rpw : 0;
    return QD_OK;
}

int qd_plan_set_timing(qd_plan *p, int enabled) {
    if (!p) return fail(QD_ERR_INVALID, "plan is NULL");
    p->timing = enabled != 0;
    if (p->cmp_a) { p->cmp_a->timing = p->timing; p->cmp_b->timing = p->timing; }
    return QD_OK;
}

int qd_plan_last_kernel_ms(qd_plan *p, float *ms) {
    if (!p || !ms) return fail(QD_ERR_INVALID, "NULL argument");
    if (p->cmp_a) {
        float a = 0.f, b = 0.f;
        int rc = qd_plan_last_kernel_ms(p->cmp_a, &a);
        if (rc == QD_OK) rc = qd_plan_last_kernel_ms(p->cmp_b, &b);
        *ms = a + b;
        return rc;
    }
    if (!p->ev_recorded) return fail(QD_ERR_INVALID, "no timed run recorded");
    HIPCHK(hipEventSynchronize(p->ev1));
    HIPCHK(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return QD_OK;
}

namespace {
// Pageable -> pinned staging copy on several host threads: one thread moves ~10-15 GB/s, which would cap the
// host-resident path far below PCIe (qd_plan_options.copy_threads; default: up to 8).
void par_memcpy(void *dst, const void *src, size_t n, unsigned n_thr) {
    if (n_thr == 0) {
        unsigned hw = std::thread::hardware_concurrency();
        n_thr = hw / 2; if (n_thr < 1) n_thr = 1; if (n_thr > 8) n_thr = 8;
    }
    if (n_thr <= 1 || n < (8u << 20)) { memcpy(dst, src, n); return; }
    const size_t slice = ((n / n_thr) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned i = 1; i < n_thr; ++i) {
        const size_t off = i * slice;
        if (off >= n) break;
        const size_t len = off + slice > n ? n - off : slice;
        th.emplace_back([=] { memcpy(static_cast<uint8_t *>(dst) + off, static_cast<const uint8_t *>(src) + off, len); });
    }
    memcpy(dst, src, slice < n ? slice : n);
    for (auto &t : th) t.join();
}

double now_ms() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

bool host_kind(int m) { return m == QD_MEM_HOST || m == QD_MEM_HOST_PINNED; }

// Host-resident stream: chunked, double-buffered H2D / kernel / D2H on two streams (slot = chunk parity).  A pageable
// buffer (QD_MEM_HOST) is staged through a pinned ring with a multi-threaded memcpy; QD_MEM_HOST_PINNED memory is the
// DMA source / target itself.  Each slot owns its device buffers AND its NCO row table, so nothing a kernel in flight on
// the other slot reads is ever touched.  Windows are kernel windows (sub-blocks for QD_EPI_CF32_BLOCKS).
int run_host(qd_plan *p, const void *src, int src_mem, uint64_t src_first, uint64_t src_count, uint64_t first_window,
             uint64_t n_windows, void *out, int out_mem, uint64_t obw) {
    const double t_begin = now_ms();
    p->stats = qd_plan_stats{};
    const int bps = bps_of(p->d.format);
    const uint64_t step = (uint64_t)p->S * p->D, rpw = (uint64_t)p->W * p->D + p->T;
    const uint64_t target_bytes = p->opt.chunk_bytes ? p->opt.chunk_bytes : (64ull << 20);
    uint64_t cw = target_bytes / (step * bps ? step * bps : 1);
    if (cw < p->geo.G) cw = p->geo.G;
    cw = (cw / p->geo.G) * p->geo.G;
    if ((p->kflags & (kGeoFastP1 | kGeoPipe3 | kGeoSpark)) && (p->jit_fn || p->fixed || p->spark)) {
        // row-aligned kernels: a launch whose first window is off the row grid goes to the per-sample kernel (launch_chain), so
        // chunks start on windows that are multiples of lcm(G, ROW / gcd(ROW, S D))
        const uint64_t ROW = (uint64_t)p->nt * spl_of(p->d.format);
        uint64_t a = ROW, b = step % ROW; while (b) { const uint64_t t = a % b; a = b; b = t; }
        const uint64_t wa = ROW / a;                                    // windows per row-grid period
        uint64_t g = p->geo.G, h = wa; while (h) { const uint64_t t = g % h; g = h; h = t; }
        uint64_t unit = (uint64_t)p->geo.G / g * wa;                    // lcm
        uint64_t grid = wa;
        if (p->spark_R > 1) {                                           // interleaved launches: chunks start where every launch's first window sits on its grid
            uint64_t a2 = unit, b2 = p->phase_unit; while (b2) { const uint64_t t = a2 % b2; a2 = b2; b2 = t; }
            unit = unit / a2 * p->phase_unit;
            grid = p->phase_unit;
        }
        if (first_window % grid == 0 && cw >= unit) cw = (cw / unit) * unit;
    }
    if (cw > n_windows) cw = n_windows ? n_windows : 1;
    const size_t in_bytes = (size_t)(((cw - 1) * step + rpw + 8) * bps), ob = (size_t)(cw * obw);
    const bool stage_in = src_mem == QD_MEM_HOST, stage_out = out_mem == QD_MEM_HOST;
    if (in_bytes > p->stage_in_bytes || ob > p->stage_out_bytes || (stage_in && in_bytes > p->pin_in_bytes) || (stage_out && ob > p->pin_out_bytes)) {
        free_streaming(p);
        for (int i = 0; i < 2; ++i) {
            if (stage_in) HIPCHK(hipHostMalloc(&p->pin_in[i], in_bytes, hipHostMallocDefault));
            if (stage_out) HIPCHK(hipHostMalloc(&p->pin_out[i], ob, hipHostMallocDefault));
            HIPCHK(hipMalloc(&p->dev_in[i], in_bytes));
            HIPCHK(hipMalloc(&p->dev_out[i], ob));
            HIPCHK(hipStreamCreateWithFlags(&p->streams[i], hipStreamNonBlocking));
        }
        p->stage_in_bytes = in_bytes; p->stage_out_bytes = ob;
        p->pin_in_bytes = stage_in ? in_bytes : 0; p->pin_out_bytes = stage_out ? ob : 0;
    }
    double stage_ms = 0;
    struct Pending { bool live = false; uint64_t w0 = 0, nw = 0; } pend[2];
    auto drain = [&](int slot) -> int {
        if (!pend[slot].live) return QD_OK;
        HIPCHK(hipStreamSynchronize(p->streams[slot]));
        if (stage_out) {
            const double t0 = now_ms();
            par_memcpy(static_cast<uint8_t *>(out) + (pend[slot].w0 - first_window) * obw, p->pin_out[slot], pend[slot].nw * obw, p->opt.copy_threads);
            stage_ms += now_ms() - t0;
        }
        pend[slot].live = false;
        return QD_OK;
    };
    // Any error after the first enqueue leaves H2D copies, kernels and D2H copies of earlier chunks in flight — with pinned
    // buffers the D2H target is the CALLER's memory.  Quiesce both slot streams before handing the status back.
    auto quiesce = [&](int status) -> int {
        for (int i = 0; i < 2; ++i) if (p->streams[i]) (void)hipStreamSynchronize(p->streams[i]);
        return status;
    };
    int slot = 0;
    for (uint64_t w = first_window; w < first_window + n_windows; w += cw, slot ^= 1) {
        // a staged slot's pinned buffers are reused: wait for its previous chunk; a pinned-to-pinned run only needs
        // stream order (same slot = same stream), so the host runs ahead and just bounds the queue depth
        int rc = (stage_in || stage_out || ((w - first_window) / cw) % 16 >= 14) ? drain(slot) : QD_OK;
        if (rc) return quiesce(rc);
        const uint64_t nw = first_window + n_windows - w < cw ? first_window + n_windows - w : cw;
        const uint64_t s0 = w * step, cnt = (nw - 1) * step + rpw;
        // keep vector loads aligned: start the slab on a multiple of 8 samples
        uint64_t s0a = s0 & ~7ull;
        if (s0a < src_first) s0a = src_first;
        const uint64_t cnta = s0 + cnt - s0a;
        if (s0a < src_first || s0a + cnta > src_first + src_count)
            return quiesce(fail(QD_ERR_INVALID, "src slab does not cover the requested windows"));
        const uint8_t *hsrc = static_cast<const uint8_t *>(src) + (s0a - src_first) * bps;
        if (stage_in) {
            const double t0 = now_ms();
            par_memcpy(p->pin_in[slot], hsrc, cnta * bps, p->opt.copy_threads);
            stage_ms += now_ms() - t0;
            hsrc = static_cast<const uint8_t *>(p->pin_in[slot]);
        }
        if (hipError_t e = hipMemcpyAsync(p->dev_in[slot], hsrc, cnta * bps, hipMemcpyHostToDevice, p->streams[slot]); e != hipSuccess)
            return quiesce(fail(QD_ERR_HIP, "hipMemcpyAsync (H2D): %s", hipGetErrorString(e)));
        rc = launch_chain(p, &p->tabs_slot[slot], p->dev_in[slot], s0a, cnta, w, nw, w, p->dev_out[slot], p->streams[slot]);
        if (rc) return quiesce(rc);
        void *hdst = stage_out ? p->pin_out[slot] : static_cast<void *>(static_cast<uint8_t *>(out) + (w - first_window) * obw);
        if (hipError_t e = hipMemcpyAsync(hdst, p->dev_out[slot], nw * obw, hipMemcpyDeviceToHost, p->streams[slot]); e != hipSuccess)
            return quiesce(fail(QD_ERR_HIP, "hipMemcpyAsync (D2H): %s", hipGetErrorString(e)));
        pend[slot].live = true; pend[slot].w0 = w; pend[slot].nw = nw;
        p->stats.bytes_h2d += cnta * bps; p->stats.bytes_d2h += nw * obw; p->stats.chunks += 1;
    }
    int rc = drain(0);
    if (rc) return quiesce(rc);
    rc = drain(1);
    if (rc) return quiesce(rc);
    p->stats.stage_ms = stage_ms;
    p->stats.wall_ms = now_ms() - t_begin;
    return rc;
}
}  // namespace

namespace {
int grow(void **buf, size_t *have, size_t need) {
    if (need <= *have) return QD_OK;
    if (*buf) { HIPCHK(hipFree(*buf)); *buf = nullptr; *have = 0; }       // (hipFree waits for the device)
    HIPCHK(hipMalloc(buf, need));
    *have = need;
    return QD_OK;
}

// windows [w0, w0 + nw) of a two-stage plan on device buffers: A filters blocks into the carrier, B transforms them; one stream, in order
int composite_device(qd_plan *p, const void *src_d, uint64_t src_first, uint64_t src_count, uint64_t w0, uint64_t nw, void *out_d, hipStream_t st) {
    qd_plan *a = p->cmp_a, *b = p->cmp_b;
    const uint64_t W = p->W;
    int rc = grow(&p->cmp_tmp, &p->cmp_tmp_bytes, (size_t)(nw * W + 16) * 8);
    if (rc) return rc;
    // the carrier is shared by every call on this plan: a call on another stream waits for the previous one's last kernel
    if (p->cmp_used) HIPCHK(hipStreamWaitEvent(st, p->cmp_done, 0));
    rc = launch_chain(a, &a->tabs_dev, src_d, src_first, src_count, w0 * a->blk_subs, nw * a->blk_subs, w0 * a->blk_subs, p->cmp_tmp, st);
    if (rc) return rc;
    rc = launch_chain(b, &b->tabs_dev, p->cmp_tmp, w0 * W, nw * W, w0, nw, w0, out_d, st);
    if (rc) return rc;
    if (!p->cmp_done) HIPCHK(hipEventCreateWithFlags(&p->cmp_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(p->cmp_done, st));
    p->cmp_used = true;
    return QD_OK;
}

int run_composite(qd_plan *p, const void *src, int src_mem, uint64_t src_first, uint64_t src_count, uint64_t first_window, uint64_t n_windows,
                  void *out, int out_mem, hipStream_t st) {
    if (n_windows == 0) return QD_OK;
    if (src_mem == QD_MEM_DEVICE && out_mem == QD_MEM_DEVICE) return composite_device(p, src, src_first, src_count, first_window, n_windows, out, st);
    if (!host_kind(src_mem) || !host_kind(out_mem))
        return fail(QD_ERR_UNSUPPORTED, "mixed host/device buffers are not supported; use both host or both device");
    // host-resident stream: plain chunks (copy in, two launches, copy out) — this path serves windows of tens of thousands of source
    // samples, where a chunk is a handful of windows; no double buffering
    const int bps = bps_of(p->d.format);
    const uint64_t step = (uint64_t)p->W * p->D, rpw = step + p->T, obw = out_bytes_per_window(p);
    uint64_t cw = (64ull << 20) / (step * bps);
    if (cw < 1) cw = 1;
    if (cw > n_windows) cw = n_windows;
    int rc = grow(&p->cmp_in, &p->cmp_in_bytes, (size_t)(((cw - 1) * step + rpw) * bps + 64));
    if (rc == QD_OK) rc = grow(&p->cmp_out, &p->cmp_out_bytes, (size_t)(cw * obw + 64));
    if (rc) return rc;
    for (uint64_t w = first_window; w < first_window + n_windows; w += cw) {
        const uint64_t nw = std::min<uint64_t>(cw, first_window + n_windows - w);
        const uint64_t s0 = w * step, sc = (nw - 1) * step + rpw;
        if (s0 < src_first || s0 + sc > src_first + src_count)
            return fail(QD_ERR_INVALID, "src slab [%llu,+%llu) does not cover samples [%llu,+%llu) needed by windows [%llu,+%llu)", (unsigned long long)src_first,
                        (unsigned long long)src_count, (unsigned long long)s0, (unsigned long long)sc, (unsigned long long)w, (unsigned long long)nw);
        HIPCHK(hipMemcpyAsync(p->cmp_in, static_cast<const uint8_t *>(src) + (s0 - src_first) * bps, (size_t)(sc * bps), hipMemcpyHostToDevice, st));
        rc = composite_device(p, p->cmp_in, s0, sc, w, nw, p->cmp_out, st);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(static_cast<uint8_t *>(out) + (w - first_window) * obw, p->cmp_out, (size_t)(nw * obw), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return QD_OK;
}
}  // namespace

int qd_plan_run(qd_plan *p, const void *src, int src_mem, uint64_t src_first, uint64_t src_count,
                uint64_t first_window, uint64_t n_windows, void *out, int out_mem, void *stream) {
    if (!p || !src || !out) return fail(QD_ERR_INVALID, "NULL argument");
    const uint64_t subs = p->blk_subs;          // 1 except QD_EPI_CF32_BLOCKS (API windows are whole blocks)
    if (first_window + n_windows > p->n_windows)
        return fail(QD_ERR_SHORT, "windows [%llu,+%llu) exceed the sink's loop (%llu windows)", (unsigned long long)first_window,
                    (unsigned long long)n_windows, (unsigned long long)p->n_windows);
    if (src_first + src_count > p->d.n_samples) return fail(QD_ERR_INVALID, "src slab exceeds the stream length");
    std::lock_guard<std::mutex> lock(p->mu);
    DeviceGuard guard(p->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->cmp_a) return run_composite(p, src, src_mem, src_first, src_count, first_window, n_windows, out, out_mem, st);
    if (src_mem == QD_MEM_DEVICE && out_mem == QD_MEM_DEVICE)
        return launch_chain(p, &p->tabs_dev, src, src_first, src_count, first_window * subs, n_windows * subs, first_window * subs, out, st);
    if (!host_kind(src_mem) || !host_kind(out_mem))
        return fail(QD_ERR_UNSUPPORTED, "mixed host/device buffers are not supported; use both host or both device");
    const uint64_t obw = out_bytes_per_window(p) / subs;      // per kernel window (a sub-block for QD_EPI_CF32_BLOCKS)
    return run_host(p, src, src_mem, src_first, src_count, first_window * subs, n_windows * subs, out, out_mem, obw);
}

int qd_plan_get_stats(const qd_plan *p, qd_plan_stats *stats) {
    if (!p || !stats) return fail(QD_ERR_INVALID, "NULL argument");
    *stats = p->stats;
    return QD_OK;
}

int qd_plan_shard_info(const qd_plan *p, uint32_t shard, qd_shard_info *info) {
    if (!p || !info) return fail(QD_ERR_INVALID, "NULL argument");
    if (shard >= p->shard_info.size()) return fail(QD_ERR_INVALID, "shard %u of %zu", shard, p->shard_info.size());
    *info = p->shard_info[shard];
    return QD_OK;
}

int qd_plan_run_sharded(qd_plan *p, const void *src, int src_mem, void *out, int out_mem) {
    if (!p || !src || !out) return fail(QD_ERR_INVALID, "NULL argument");
    if (!host_kind(src_mem) || !host_kind(out_mem)) return fail(QD_ERR_INVALID, "qd_plan_run_sharded takes host buffers (QD_MEM_HOST / QD_MEM_HOST_PINNED)");
    if (p->shards.empty()) return qd_plan_run(p, src, src_mem, 0, p->d.n_samples, 0, p->n_windows, out, out_mem, nullptr);
    // one host thread per shard: each drives its device's double-buffered ring; every shard reads its windows' source
    // range — halo included — straight from the host buffer, so there is no exchange step at all
    const uint64_t obw_api = out_bytes_per_window(p);
    const size_t n = p->shards.size();
    std::vector<int> rcs(n, QD_OK);
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (size_t g = 0; g < n; ++g) {
        th.emplace_back([&, g] {
            const qd_shard_info &si = p->shard_info[g];
            if (si.w1 == si.w0) return;
            rcs[g] = qd_plan_run(p->shards[g], src, src_mem, 0, p->d.n_samples, si.w0, si.w1 - si.w0,
                                 static_cast<uint8_t *>(out) + si.w0 * obw_api, out_mem, nullptr);
            if (rcs[g]) errs[g] = g_err;         // thread-local message of the worker
        });
    }
    for (auto &t : th) t.join();
    p->stats = qd_plan_stats{};
    for (size_t g = 0; g < n; ++g) {
        if (rcs[g]) return fail(rcs[g], "shard %zu (device %d): %s", g, p->shard_info[g].device, errs[g].c_str());
        const qd_plan_stats &cs = p->shards[g]->stats;
        p->stats.wall_ms = std::max(p->stats.wall_ms, cs.wall_ms); p->stats.stage_ms += cs.stage_ms;
        p->stats.bytes_h2d += cs.bytes_h2d; p->stats.bytes_d2h += cs.bytes_d2h; p->stats.chunks += cs.chunks;
    }
    return QD_OK;
}

int qd_plan_run_sharded_device(qd_plan *p, void *const *slabs, void *const *outs, int sync) {
    if (!p || !slabs || !outs) return fail(QD_ERR_INVALID, "NULL argument");
    if (p->cmp_a) return fail(QD_ERR_UNSUPPORTED, "a two-stage plan (window larger than the LDS tile) has no pre-split device path");
    const size_t n = p->shard_info.size();
    const int bps = bps_of(p->d.format);
    std::vector<qd_plan *> plans(n, p);
    for (size_t g = 0; g < n && !p->shards.empty(); ++g) plans[g] = p->shards[g];
    for (size_t g = 0; g < n; ++g) {
        const qd_shard_info &si = p->shard_info[g];
        if (si.w1 == si.w0) continue;
        if (!slabs[g] || !outs[g]) return fail(QD_ERR_INVALID, "shard %zu: NULL slab / out", g);
        if (si.halo && (g + 1 >= n || p->shard_info[g + 1].own_count < si.halo))
            return fail(QD_ERR_INVALID, "shard %zu needs a %llu-sample halo its neighbour does not own: use fewer shards", g, (unsigned long long)si.halo);
    }
    for (size_t g = 0; g < n; ++g) {
        const qd_shard_info &si = p->shard_info[g];
        if (si.w1 == si.w0) continue;
        qd_plan *c = plans[g];
        std::lock_guard<std::mutex> lock(c->mu);
        DeviceGuard guard(si.device);
        if (!c->streams[0]) HIPCHK(hipStreamCreateWithFlags(&c->streams[0], hipStreamNonBlocking));
        if (si.halo)     // the one exchange step of a pre-split device-resident stream: (W-S)*D+T samples from the next slab
            HIPCHK(hipMemcpyPeerAsync(static_cast<uint8_t *>(slabs[g]) + si.own_count * bps, si.device, slabs[g + 1],
                                      p->shard_info[g + 1].device, si.halo * bps, c->streams[0]));
        const uint64_t subs = c->blk_subs;
        int rc = launch_chain(c, &c->tabs_dev, slabs[g], si.own_first, si.own_count + si.halo, si.w0 * subs, (si.w1 - si.w0) * subs,
                              si.w0 * subs, outs[g], c->streams[0]);
        if (rc) return rc;
    }
    if (sync) {
        for (size_t g = 0; g < n; ++g) {
            if (p->shard_info[g].w1 == p->shard_info[g].w0 || !plans[g]->streams[0]) continue;
            DeviceGuard guard(p->shard_info[g].device);
            HIPCHK(hipStreamSynchronize(plans[g]->streams[0]));
        }
    }
    return QD_OK;
}

int qd_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(QD_ERR_INVALID, "NULL argument");
    *ptr = nullptr;
    if (bytes == 0) return QD_OK;
    HIPCHK(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return QD_OK;
}
int qd_host_free(void *ptr) { if (ptr) HIPCHK(hipHostFree(ptr)); return QD_OK; }
int qd_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(QD_ERR_INVALID, "NULL / empty range");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return QD_OK;
}
int qd_host_unregister(void *ptr) { if (ptr) HIPCHK(hipHostUnregister(ptr)); return QD_OK; }

// ------------------------------------------------------------------ fine-grained ops
//
// These mirror one `read_at` each, so a host may call them once per window: no hipMalloc /
// hipFree / plan construction per call.  Device temporaries come from a process-wide pool of grow-only workspaces; a
// workspace is handed to one call at a time and remembers the stream that used it last, so reuse on the same stream
// needs no synchronisation (stream order) and reuse on another stream waits for the old one first.  The FFT-based
// calls keep their plans in a small cache.  All launches go to the calling thread's stream (qd_set_stream).

namespace {

thread_local hipStream_t g_stream = nullptr;

struct Workspace {
    static constexpr int kSlots = 6;
    void *buf[kSlots] = {};
    size_t cap[kSlots] = {};
    int device = 0;
    hipStream_t last = nullptr;          // compared, never dereferenced: the caller may have destroyed it since
    hipEvent_t done = nullptr;           // recorded behind the last call that used the buffers
    bool used = false;
    int get(int i, size_t bytes, void **out) {
        if (bytes > cap[i]) {
            if (buf[i]) {
                if (used && done) HIPCHK(hipEventSynchronize(done));      // earlier calls; this call has not touched slot i yet
                HIPCHK(hipFree(buf[i]));
                buf[i] = nullptr; cap[i] = 0;
            }
            const size_t want = (bytes + bytes / 4 + 4095) & ~(size_t)4095;
            HIPCHK(hipMalloc(&buf[i], want));
            cap[i] = want;
        }
        *out = buf[i];
        return QD_OK;
    }
    void release_buffers() {
        for (int i = 0; i < kSlots; ++i) { if (buf[i]) (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
        if (done) (void)hipEventDestroy(done);
        done = nullptr;
    }
};

std::mutex g_ws_mu;
std::vector<Workspace *> g_ws_idle;

struct WsLease {                        // one workspace for the duration of a call
    Workspace *ws = nullptr;
    hipStream_t st_ = nullptr;
    int rc = QD_OK;
    explicit WsLease(hipStream_t st) : st_(st) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lock(g_ws_mu);
            size_t pick = g_ws_idle.size();
            for (size_t i = 0; i < g_ws_idle.size(); ++i) {
                if (g_ws_idle[i]->device != dev) continue;
                if (pick == g_ws_idle.size() || g_ws_idle[i]->last == st) pick = i;
                if (g_ws_idle[i]->last == st) break;
            }
            if (pick < g_ws_idle.size()) { ws = g_ws_idle[pick]; g_ws_idle.erase(g_ws_idle.begin() + pick); }
        }
        if (!ws) { ws = new Workspace(); ws->device = dev; }
        // hand-over between streams: the new stream waits (on the device) for the event behind the workspace's last call
        // (always: `last` is kept to prefer a workspace this stream used, never to skip the wait — a handle may be a new stream at a recycled address)
        if (ws->used && ws->done && hipStreamWaitEvent(st, ws->done, 0) != hipSuccess) rc = fail(QD_ERR_HIP, "workspace hand-over: hipStreamWaitEvent failed");
        if (!ws->done && hipEventCreateWithFlags(&ws->done, hipEventDisableTiming) != hipSuccess) { ws->done = nullptr; rc = fail(QD_ERR_HIP, "workspace: hipEventCreate failed"); }
        ws->last = st; ws->used = true;
    }
    ~WsLease() {
        if (ws->done) (void)hipEventRecord(ws->done, st_);         // whatever this call enqueued (also on an error return)
        std::lock_guard<std::mutex> lock(g_ws_mu);
        g_ws_idle.push_back(ws);
    }
    int get(int i, size_t bytes, void **out) { return ws->get(i, bytes, out); }
};

// plans of the FFT-based fine-grained calls, keyed by (device, width, stride, kind)
// Entries are handed out as shared_ptr: an eviction (cache full) or qd_release_workspaces only drops the CACHE's reference,
// the plan is destroyed when the last caller still using it lets go — nobody locks a freed mutex or launches on a freed plan.
struct CachedPlan {
    int device; uint64_t W, S; int kind; qd_plan *plan; std::mutex mu;
    CachedPlan(int dev, uint64_t w, uint64_t s, int k, qd_plan *p) : device(dev), W(w), S(s), kind(k), plan(p) {}
    ~CachedPlan() { if (plan) { DeviceGuard guard(device); (void)qd_plan_destroy(plan); } }
};
std::mutex g_pc_mu;
std::vector<std::shared_ptr<CachedPlan>> g_plan_cache;
constexpr size_t kPlanCacheMax = 32;
constexpr uint64_t kOpenEnded = 1ull << 40;      // "any number of windows": the cached plans size no loop from it

int cached_fft_plan(uint64_t W, uint64_t S, int kind, std::shared_ptr<CachedPlan> *out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_pc_mu);
    for (const std::shared_ptr<CachedPlan> &c : g_plan_cache)
        if (c->device == dev && c->W == W && c->S == S && c->kind == kind) { *out = c; return QD_OK; }
    qd_chain_desc d{};
    d.struct_size = sizeof d;
    d.format = QD_FMT_CF32; d.sample_rate = 1;
    d.n_samples = kOpenEnded;
    d.width = W; d.stride = S; d.epilogue = QD_EPI_NORMS_F32;
    qd_plan_options o{};
    o.struct_size = sizeof o;
    o.kernel_policy = QD_KERNEL_NO_PLAN_TIME;      // a per-window helper must not stall on a compile
    qd_plan *p = nullptr;
    int rc = qd_plan_create_ex(&d, &o, &p);
    if (rc) return rc;
    if (kind == 1) {                               // take_fft: one irregular row per tile, per-sample (unaligned) kernel
        p->geo.G = 1;
        uint32_t raw_elems = 0;
        p->geo.lds_bytes = lds_for(1, p->W, p->S, p->D, p->T, &raw_elems, 1, 1, false);
        p->geo.lds_main = p->geo.lds_bytes;
        p->geo.lds_raw_elems = raw_elems;
        p->fn = p->fn_unaligned;
        p->spark = false; p->kflags = 0;           // irregular rows: the per-sample generic kernel only
    }
    if (g_plan_cache.size() >= kPlanCacheMax) g_plan_cache.erase(g_plan_cache.begin());     // the oldest entry; destroyed once unused
    g_plan_cache.push_back(std::make_shared<CachedPlan>(dev, W, S, kind, p));
    *out = g_plan_cache.back();
    return QD_OK;
}

// ---- Bluestein tables per width (host f64 arithmetic, rounded once to f32), cached per device
struct BluesteinTab {
    int device = 0; uint32_t W = 0, M = 0, logM = 0; double2 *chirp = nullptr, *Bbr = nullptr, *tw = nullptr;
    BluesteinTab() = default;
    BluesteinTab(const BluesteinTab &) = delete;
    BluesteinTab &operator=(const BluesteinTab &) = delete;
    // hipFree waits for the device: a kernel still reading the tables (its caller has already let go) finishes first
    ~BluesteinTab() { DeviceGuard guard(device); if (chirp) (void)hipFree(chirp); if (Bbr) (void)hipFree(Bbr); if (tw) (void)hipFree(tw); }
};
std::mutex g_bt_mu;
std::vector<std::shared_ptr<BluesteinTab>> g_bt;       // by reference count, like the plan cache: eviction never frees under a caller

void fft64_inplace(std::vector<double> &re, std::vector<double> &im, uint32_t logM) {   // radix-2 DIT, natural order out
    const uint32_t M = 1u << logM;
    for (uint32_t i = 0; i < M; ++i) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < logM; ++b) if (i & (1u << b)) r |= 1u << (logM - 1 - b);
        if (r > i) { std::swap(re[i], re[r]); std::swap(im[i], im[r]); }
    }
    for (uint32_t h = 1; h < M; h <<= 1) {
        for (uint32_t j = 0; j < h; ++j) {
            const double ang = -kPi64 * (double)j / (double)h, wr = std::cos(ang), wi = std::sin(ang);
            for (uint32_t i = j; i < M; i += 2 * h) {
                const double vr = re[i + h] * wr - im[i + h] * wi, vi = re[i + h] * wi + im[i + h] * wr;
                re[i + h] = re[i] - vr; im[i + h] = im[i] - vi;
                re[i] += vr; im[i] += vi;
            }
        }
    }
}

int bluestein_tab(uint32_t W, std::shared_ptr<BluesteinTab> *out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_bt_mu);
    for (const std::shared_ptr<BluesteinTab> &e : g_bt) if (e->device == dev && e->W == W) { *out = e; return QD_OK; }
    std::shared_ptr<BluesteinTab> tp = std::make_shared<BluesteinTab>();
    BluesteinTab &t = *tp;
    t.device = dev; t.W = W;
    t.logM = ilog2(2ull * W - 1); t.M = 1u << t.logM;
    const uint32_t M = t.M;
    std::vector<double2> chirp(W), Bbr(M), tw(M / 2 ? M / 2 : 1);
    std::vector<double> br(M, 0.0), bi(M, 0.0);
    for (uint32_t n = 0; n < W; ++n) {
        const uint64_t q = ((uint64_t)n * n) % (2ull * W);       // n^2 mod 2W: the angle is reduced exactly, in integers
        const double ang = kPi64 * (double)q / (double)W;
        const double cr = std::cos(ang), ci = std::sin(ang);     // b[n] = e^{+i pi n^2 / W}
        chirp[n] = make_double2(cr, -ci);                         // c[n] = conj(b[n])
        br[n] = cr; bi[n] = ci;
        if (n) { br[M - n] = cr; bi[M - n] = ci; }
    }
    fft64_inplace(br, bi, t.logM);
    for (uint32_t r = 0; r < M; ++r) {
        uint32_t k = 0;
        for (uint32_t b = 0; b < t.logM; ++b) if (r & (1u << b)) k |= 1u << (t.logM - 1 - b);
        Bbr[r] = make_double2(br[k] / (double)M, bi[k] / (double)M);
    }
    for (uint32_t k = 0; k < M / 2; ++k) {
        const double ang = -2.0 * kPi64 * (double)k / (double)M;
        tw[k] = make_double2(std::cos(ang), std::sin(ang));
    }
    if (M / 2 == 0) tw[0] = make_double2(1.0, 0.0);
    HIPCHK(hipMalloc(&t.chirp, chirp.size() * 16)); HIPCHK(hipMalloc(&t.Bbr, Bbr.size() * 16)); HIPCHK(hipMalloc(&t.tw, tw.size() * 16));
    HIPCHK(hipMemcpy(t.chirp, chirp.data(), chirp.size() * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t.Bbr, Bbr.data(), Bbr.size() * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t.tw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice));
    if (g_bt.size() >= 64) g_bt.erase(g_bt.begin());               // the width slider walks through many lengths: bound the cache
    g_bt.push_back(tp);
    *out = tp;
    return QD_OK;
}

int bluestein_rows(const float2 *src, uint64_t in_first, const uint64_t *offs_d, const float *win_d, size_t W, size_t n_rows,
                   float *dst, hipStream_t st) {
    if (W > 4096) return fail(QD_ERR_UNSUPPORTED, "take_fft width %zu: widths that are not a power of two are built up to 4096 (the reference front end's slider range, src/eui/mod.rs:157)", W);
    std::shared_ptr<BluesteinTab> tp;                  // held until the launch is enqueued
    int rc = bluestein_tab((uint32_t)W, &tp);
    if (rc) return rc;
    const BluesteinTab &t = *tp;
    if ((size_t)t.M * 16 > 48 * 1024)     // per device, cheap: raise the dynamic-LDS limit to the hardware maximum
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bluestein), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
    hipLaunchKernelGGL(k_bluestein, dim3((uint32_t)n_rows), dim3(256), (size_t)t.M * 16, st, src, in_first, offs_d, win_d, (uint32_t)W, t.M, t.logM,
                       t.chirp, t.Bbr, t.tw, dst);
    HIPCHK(hipGetLastError());
    return QD_OK;
}

int finish_call(int mem, hipStream_t st) {         // host buffers: results must be there on return
    if (mem != QD_MEM_DEVICE) HIPCHK(hipStreamSynchronize(st));
    return QD_OK;
}

}  // namespace

int qd_set_stream(void *stream) { g_stream = static_cast<hipStream_t>(stream); return QD_OK; }

int qd_release_workspaces(void) {
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        for (Workspace *w : g_ws_idle) {
            DeviceGuard guard(w->device);
            if (w->used && w->done) (void)hipEventSynchronize(w->done);
            w->release_buffers();
            delete w;
        }
        g_ws_idle.clear();
    }
    std::vector<std::shared_ptr<CachedPlan>> dropped;
    {
        std::lock_guard<std::mutex> lock(g_pc_mu);
        dropped.swap(g_plan_cache);
    }
    dropped.clear();                               // plans nobody is running are destroyed here, the others when their call returns
    {
        std::lock_guard<std::mutex> lock(g_bt_mu);
        g_bt.clear();
    }
    return QD_OK;
}

int qd_unpack(int fmt, const void *bytes, size_t n_pairs, qd_c32 *out, int mem) {
    if (fmt < 0 || fmt > 3) return fail(QD_ERR_INVALID, "unknown format %d", fmt);
    if (n_pairs == 0) return QD_OK;
    if (!bytes || !out) return fail(QD_ERR_INVALID, "NULL buffer");
    const size_t ib = n_pairs * qd_pair_bytes(fmt), ob = n_pairs * 8;
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    const void *src = bytes; void *dst = out;
    if (mem != QD_MEM_DEVICE) {
        void *di = nullptr, *dout = nullptr;
        int rc = ws.get(0, ib, &di); if (rc) return rc;
        rc = ws.get(1, ob, &dout); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(di, bytes, ib, hipMemcpyHostToDevice, st));
        src = di; dst = dout;
    }
    size_t blocks = (n_pairs + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_unpack, dim3((uint32_t)blocks), dim3(256), 0, st, fmt, static_cast<const uint8_t *>(src), n_pairs,
                       static_cast<float2 *>(dst));
    HIPCHK(hipGetLastError());
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(out, dst, ob, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

int qd_shift(qd_c32 *buf, size_t n, uint64_t abs_off, double ratio, int mem) {
    if (n == 0) return QD_OK;
    if (!buf) return fail(QD_ERR_INVALID, "NULL buffer");
    constexpr uint32_t ROW = 512;
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    float2 *d = reinterpret_cast<float2 *>(buf);
    if (mem != QD_MEM_DEVICE) {
        void *db = nullptr;
        int rc = ws.get(0, n * 8, &db); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(db, buf, n * 8, hipMemcpyHostToDevice, st));
        d = static_cast<float2 *>(db);
    }
    const uint64_t r0 = abs_off / ROW, r1 = (abs_off + n + ROW - 1) / ROW, rows = r1 - r0;
    void *rt = nullptr, *jt = nullptr;
    int rc = ws.get(2, rows * sizeof(RowBase), &rt); if (rc) return rc;
    rc = ws.get(3, ROW * sizeof(double2), &jt); if (rc) return rc;
    hipLaunchKernelGGL(k_rowtab, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, st, ratio, ROW, r0, rows, (uint64_t)0, static_cast<RowBase *>(rt));
    hipLaunchKernelGGL(k_jtab, dim3(2), dim3(256), 0, st, ratio, ROW, static_cast<double2 *>(jt));
    const int so = (std::fabs(ratio) * (double)(abs_off + n) > 268435456.0) ? 1 : 0;
    const uint32_t grid = (uint32_t)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(k_shift, dim3(grid), dim3(256), 0, st, d, abs_off, (uint64_t)n, ratio, static_cast<const RowBase *>(rt), r0, rows,
                       static_cast<const double2 *>(jt), so);
    HIPCHK(hipGetLastError());
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(buf, d, n * 8, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

int qd_lowpass_block(const float *taps, size_t T, uint64_t D, const qd_c32 *raw, size_t valid, qd_c32 *out,
                     size_t out_cap, size_t *produced, int mem) {
    if (!taps || !raw || !out || !produced) return fail(QD_ERR_INVALID, "NULL argument");
    if (T < 2 || D == 0) return fail(QD_ERR_PANIC, "size < 2 or decimate 0 (src/filter.rs:47,74)");
    if (valid < T) return fail(QD_ERR_PANIC, "valid < filter.len(): usize underflow at src/filter.rs:76");
    size_t out_n = (size_t)((uint64_t)(valid - T) / D);
    if (out_n > out_cap) return fail(QD_ERR_PANIC, "buf too small for %zu outputs (src/filter.rs:78-80)", out_n);
    *produced = out_n;
    if (out_n == 0) return QD_OK;
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    void *dt = nullptr;
    int rc = ws.get(2, T * 4, &dt); if (rc) return rc;
    // taps are always host memory (O(T)); the caller may reuse its array at once, so this small copy is synchronous
    HIPCHK(hipMemcpyAsync(dt, taps, T * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    const float2 *r = reinterpret_cast<const float2 *>(raw);
    float2 *o = reinterpret_cast<float2 *>(out);
    if (mem != QD_MEM_DEVICE) {
        void *dr = nullptr, *dout = nullptr;
        rc = ws.get(0, valid * 8, &dr); if (rc) return rc;
        rc = ws.get(1, out_n * 8, &dout); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(dr, raw, valid * 8, hipMemcpyHostToDevice, st));
        r = static_cast<const float2 *>(dr); o = static_cast<float2 *>(dout);
    }
    size_t blocks = (out_n + 127) / 128; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_lowpass_block, dim3((uint32_t)blocks), dim3(128), 0, st, static_cast<const float *>(dt), (uint32_t)T, D, r,
                       (uint64_t)valid, o, (uint64_t)out_n);
    HIPCHK(hipGetLastError());
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(out, o, out_n * 8, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

int qd_fft_norm_batch(const qd_c32 *in, size_t W, size_t n_fft, size_t in_stride, float *norms, int mem) {
    if (n_fft == 0) return QD_OK;
    if (!in || !norms) return fail(QD_ERR_INVALID, "NULL buffer");
    if (in_stride == 0) return fail(QD_ERR_INVALID, "in_stride 0");
    if (!is_pow2(W)) return fail(QD_ERR_PANIC, "Radix4 requires a power-of-two width (rustfft API contract), got %zu", W);
    std::shared_ptr<CachedPlan> c;
    int rc = cached_fft_plan(W, in_stride, 0, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    qd_plan *p = c->plan;
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    const uint64_t have = (uint64_t)(n_fft - 1) * in_stride + W;
    const void *src = in; void *dst = norms;
    if (mem != QD_MEM_DEVICE) {
        void *di = nullptr, *dout = nullptr;
        rc = ws.get(0, have * 8, &di); if (rc) return rc;
        rc = ws.get(1, n_fft * W * 4, &dout); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(di, in, have * 8, hipMemcpyHostToDevice, st));
        src = di; dst = dout;
    }
    rc = launch_chain(p, &p->tabs_dev, src, 0, have, 0, n_fft, 0, dst, st);
    if (rc) return rc;
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(norms, dst, n_fft * W * 4, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

int qd_take_fft(const qd_c32 *in, uint64_t in_first, size_t n_in, uint64_t samples_len, int has_slice,
                uint64_t start, uint64_t end, size_t W, int windowing, size_t output_len, float *rows, int mem) {
    if (!in || !rows) return fail(QD_ERR_INVALID, "NULL buffer");
    if (W < 1 || W > (1u << 20)) return fail(QD_ERR_UNSUPPORTED, "take_fft width %zu", W);
    if (!has_slice) {                                                             // src/ffts.rs:27-30
        if (samples_len < W) return fail(QD_ERR_PANIC, "len < width underflows (src/ffts.rs:29)");
        start = 0; end = samples_len - W;
    }
    if (!(end > start)) return fail(QD_ERR_PANIC, "Invalid slice: end (%llu) must be greater than start (%llu)", (unsigned long long)end, (unsigned long long)start);
    if (!(end < samples_len)) return fail(QD_ERR_PANIC, "Slice end (%llu) exceeds sample length (%llu)", (unsigned long long)end, (unsigned long long)samples_len);
    const uint64_t visible = end - start;
    if (!(visible > output_len)) return fail(QD_ERR_INVALID, "Visible samples (%llu) must be greater than output length (%zu)", (unsigned long long)visible, output_len);
    if (output_len == 0) return QD_OK;
    // row offsets exactly as the reference forms them (f64 step, round half away from zero, saturating cast)
    const double step = (double)visible / (double)output_len;                     // :50
    std::vector<uint64_t> offs(output_len);
    for (size_t i = 0; i < output_len; ++i) {
        double r = std::round(step * (double)i);
        uint64_t ri = !(r > 0) ? 0 : (r >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)r);
        offs[i] = start + ri;                                                     // :60
        if (offs[i] < in_first || offs[i] + W > in_first + n_in || offs[i] + W > samples_len)
            return fail(QD_ERR_SHORT, "row %zu at sample %llu is not inside the provided block / the stream (read_exact_at, src/ffts.rs:62)", i, (unsigned long long)offs[i]);
    }
    std::vector<float> win;
    if (windowing == 1) {                                                         // generate_blackman_harris_window, :110-119
        win.resize(W);
        const float tau = 6.28318530717958647692528676655900577f;
        for (size_t i = 0; i < W; ++i) {
            float x = tau * (float)i / (float)(W - 1);
            win[i] = 0.35875f - 0.48829f * std::cos(x) + 0.14128f * std::cos(2.0f * x) - 0.01168f * std::cos(3.0f * x);
        }
    }
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    void *doffs = nullptr, *dwin = nullptr;
    int rc = ws.get(2, output_len * 8, &doffs); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(doffs, offs.data(), output_len * 8, hipMemcpyHostToDevice, st));
    if (!win.empty()) {
        rc = ws.get(3, W * 4, &dwin); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(dwin, win.data(), W * 4, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipStreamSynchronize(st));               // offs / win are locals of this call
    const void *src = in; void *dst = rows;
    if (mem != QD_MEM_DEVICE) {
        void *di = nullptr, *dout = nullptr;
        rc = ws.get(0, n_in * 8, &di); if (rc) return rc;
        rc = ws.get(1, output_len * W * 4, &dout); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(di, in, n_in * 8, hipMemcpyHostToDevice, st));
        src = di; dst = dout;
    }
    if (is_pow2(W)) {
        // rustfft's planner gives a power-of-two length to its Radix4 as well: the chain kernel's FFT (bit-exact against
        // the oracle's restatement), rows gathered at irregular offsets by the per-sample kernel
        std::shared_ptr<CachedPlan> c;
        rc = cached_fft_plan(W, W, 1, &c);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(c->mu);
        qd_plan *p = c->plan;
        p->row_offsets_d = static_cast<const uint64_t *>(doffs);
        p->window_d = static_cast<const float *>(dwin);
        p->d.n_samples = in_first + n_in;
        rc = launch_chain(p, &p->tabs_dev, src, in_first, n_in, 0, output_len, 0, dst, st);
        p->row_offsets_d = nullptr; p->window_d = nullptr;
    } else {
        rc = bluestein_rows(static_cast<const float2 *>(src), in_first, static_cast<const uint64_t *>(doffs), static_cast<const float *>(dwin),
                            W, output_len, static_cast<float *>(dst), st);
    }
    if (rc) return rc;
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(rows, dst, output_len * W * 4, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

int qd_device_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(QD_ERR_INVALID, "NULL argument");
    *ptr = nullptr;
    if (bytes == 0) return QD_OK;
    HIPCHK(hipMalloc(ptr, bytes));
    return QD_OK;
}

int qd_device_free(void *ptr) {
    if (ptr) HIPCHK(hipFree(ptr));
    return QD_OK;
}

int qd_device_copy(void *dst, int dst_mem, const void *src, int src_mem, size_t bytes) {
    if (bytes == 0) return QD_OK;
    if (!dst || !src) return fail(QD_ERR_INVALID, "NULL buffer");
    if ((dst_mem != QD_MEM_HOST && dst_mem != QD_MEM_DEVICE) || (src_mem != QD_MEM_HOST && src_mem != QD_MEM_DEVICE))
        return fail(QD_ERR_INVALID, "unknown memory kind");
    const hipMemcpyKind kind = dst_mem == QD_MEM_DEVICE ? (src_mem == QD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice)
                                                        : (src_mem == QD_MEM_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
    HIPCHK(hipMemcpy(dst, src, bytes, kind));
    return QD_OK;
}

int qd_gen(const int64_t *cos_hz, size_t n_cos, uint64_t sample_rate, uint64_t first, size_t n, qd_c32 *out, int mem) {
    if (!cos_hz || n_cos == 0) return fail(QD_ERR_INVALID, "cos cannot be empty (src/gen.rs:18)");
    if (sample_rate == 0) return fail(QD_ERR_INVALID, "sample rate may not be zero (src/gen.rs:19)");
    if (n == 0) return QD_OK;
    if (!out) return fail(QD_ERR_INVALID, "NULL buffer");
    const hipStream_t st = g_stream;
    WsLease ws(st);
    if (ws.rc) return ws.rc;
    void *dc = nullptr;
    int rc = ws.get(2, n_cos * 8, &dc); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dc, cos_hz, n_cos * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));              // the tone list is the caller's array
    float2 *o = reinterpret_cast<float2 *>(out);
    if (mem != QD_MEM_DEVICE) { void *dout = nullptr; rc = ws.get(1, n * 8, &dout); if (rc) return rc; o = static_cast<float2 *>(dout); }
    size_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_gen, dim3((uint32_t)blocks), dim3(256), 0, st, static_cast<const int64_t *>(dc), (uint32_t)n_cos, sample_rate, first, n, o);
    HIPCHK(hipGetLastError());
    if (mem != QD_MEM_DEVICE) HIPCHK(hipMemcpyAsync(out, o, n * 8, hipMemcpyDeviceToHost, st));
    return finish_call(mem, st);
}

}  // extern "C"
