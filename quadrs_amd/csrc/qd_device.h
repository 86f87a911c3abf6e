// qd_device.h — device-side arithmetic shared by every kernel of the engine (gfx950 only).
//
// Everything here is written so that, compiled with -ffp-contract=off, each f32 operation
// rounds exactly where the reference's Rust code rounds (no FMA contraction anywhere on
// the f32 data path).  f64 fma() calls in the NCO are explicit and intentional.
#pragma once

#if !defined(__HIPCC_RTC__)      // hiprtc (plan-time specialisation) supplies the HIP builtins itself
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned char uint8_t; typedef signed char int8_t; typedef unsigned short uint16_t; typedef short int16_t;
typedef unsigned int uint32_t; typedef int int32_t; typedef unsigned long long uint64_t; typedef long long int64_t;
typedef unsigned long uintptr_t; typedef unsigned long size_t;
#endif

namespace qd {

// ---------------------------------------------------------------- complex f32 (num-complex 0.4.6)

// Complex<f32> * Complex<f32>: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
// reference: src/shift.rs:51 (`buf[i] *= mul`), rustfft twiddle products.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    float2 r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
// The same product as three packed instructions: (a.re*b.re, a.re*b.im), (a.im*b.im, a.im*b.re), then (lo - lo, hi + hi).
// Every multiply and the final add / subtract round exactly as in cmul; hipcc's own packing of cmul needs seven VALU
// instructions (a register copy to splat a.im, two packed adds of which half the lanes are discarded, two more copies).
__device__ __forceinline__ float2 cmul_pk(float2 a, float2 b) {
    typedef float v2f_t __attribute__((ext_vector_type(2)));
    const v2f_t av = {a.x, a.y}, bv = {b.x, b.y};
    v2f_t r, t;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,0]"
        : "=&v"(r), "=&v"(t) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// rustfft twiddles::rotate_90, Forward: (im, -re)
__device__ __forceinline__ float2 rot90(float2 v) { return make_float2(v.y, -v.x); }

// num-complex norm() = re.hypot(im) -> glibc 2.35 __hypotf, which is
// (float)sqrt((double)x*x + (double)y*y) with inf/nan screened first (checked bit-for-bit
// against glibc on 2e8 random pairs).  reference: src/fft.rs:53,95-96.
__device__ __forceinline__ float norm_ieee(float x, float y) {      // the literal form: IEEE f64 sqrt, ~20 f64-rate instructions
    if (!(__builtin_isfinite(x) && __builtin_isfinite(y))) {
        if (__builtin_isinf(x) || __builtin_isinf(y)) return __builtin_inff();
        return x + y;
    }
    const double dx = (double)x, dy = (double)y;
    return (float)__builtin_sqrt(__builtin_fma(dx, dx, dy * dy));
}
// Round 4: the same VALUE in ~40 % fewer issue slots.  s = x^2 + y^2 in f64 as glibc forms it (the products are exact, one
// rounding); then r = sf * rsq(sf), sf = (float)s (v_rsq_f32: 1 ulp, so r has ~22 good bits) and ONE Newton step in f64,
// g = r + (s - r^2) * (rsq / 2): |g - sqrt(s)| <= 2^-43.4 sqrt(s) (s - r^2 is exact in an fma; the dropped second-order term is
// e^2 / 8 r^3 <= 2^-45.4, the reciprocal's own error enters through the correction only: 2^-43.9).  (float)g is the reference's
// result — including the DOUBLE rounding of (float)RN64(sqrt(s)) — whenever no f32 rounding boundary lies within that distance of
// g, because then sqrt(s), RN64(sqrt(s)) and g all sit on the same side of every boundary.  A boundary is an f64 whose low 29
// mantissa bits are 0x10000000 (the overflow threshold included); g is "near" one when those bits are within 2^12 f64-ulps of it
// (five times the bound), which happens for 3e-5 of random inputs: those lanes, and everything with s outside [2^-96, 2^96)
// (zeros, subnormal results, overflow, inf / nan: sf or the reciprocal would leave the normal f32 range), take the IEEE form.
// The test tree restates this on the CPU (with the reciprocal perturbed by +-2 ulp) against glibc's hypotf on 10^9 pairs and
// every tie a Pythagorean triple can make (tests/test_oracle_golden.py::test_device_norm_model_equals_hypotf); on the device
// tests/test_gpu_parity.py::test_norm_equals_hypotf runs 2.7e8 pairs + the structured cases through the kernels.
// norm_fast: the short form alone; `slow` says the IEEE form must decide (callers with several values to take test ONE flag).
__device__ __forceinline__ float norm_fast(float x, float y, bool &slow) {
    const double dx = (double)x, dy = (double)y;
    const double s = __builtin_fma(dx, dx, dy * dy);      // both squares are exact in f64: the same single rounding as dx*dx + dy*dy
    const uint32_t hi = (uint32_t)(__builtin_bit_cast(uint64_t, s) >> 32);
    const float sf = (float)s;
    const float q = __builtin_amdgcn_rsqf(sf);
    const float r = sf * q, qh = 0.5f * q;
    const double rd = (double)r;
    const double e = __builtin_fma(-rd, rd, s);
    const double g = __builtin_fma(e, (double)qh, rd);
    const uint32_t m = ((uint32_t)__builtin_bit_cast(uint64_t, g) & 0x1fffffffu) - (0x10000000u - 4096u);
    slow = !((hi - ((1023u - 96u) << 20)) < (192u << 20) && m >= 8192u);
    return (float)g;
}
__device__ __forceinline__ float norm_ref(float2 v) {
    bool slow;
    const float r = norm_fast(v.x, v.y, slow);
    if (__builtin_expect(slow, 0)) return norm_ieee(v.x, v.y);
    return r;
}

// ---------------------------------------------------------------- unpack (src/lib.rs:241-255)

// Correctly rounded f / d for the small integers the sample formats produce, without the division sequence and in TWO operations:
// 1 / d = hi + lo (hi = RN(1/d), lo = RN(1/d - hi): 1/d to 2^-48), q = fma(f, hi, RN(f * lo)).  f hi + RN(f lo) is the quotient to
// 2^-47 relative and the fma rounds it ONCE; no quotient of these integers lies that close to an f32 rounding boundary unless it is
// exact (|f 2^k - d m| >= 1 for integers).  Verified against the IEEE quotient for EVERY input of the three formats (256 / 256 /
// 65536 values; tests/test_unpack_division.py restates the two operations in rational arithmetic; the GPU suite runs every code
// through the kernels).  Round 4: was q = f RN(1/d) plus a residual step, three operations.
__device__ __forceinline__ float div_small(float f, float hi, float lo) { return __builtin_fmaf(f, hi, f * lo); }
constexpr float kInv127Hi = 0x1.020408p-7f, kInv127Lo = 0x1.020408p-35f;
constexpr float kInv255Hi = 0x1.010102p-8f, kInv255Lo = -0x1.fdfdfep-33f;
constexpr float kInv65535Hi = 0x1.0001p-16f, kInv65535Lo = 0x1.0001p-48f;
// one 8-bit pair component out of a packed word: the signed byte as f32 in one instruction (v_cvt_f32_i32 with an SDWA byte select and
// sign extension), and hipcc pairs the two components of a sample into v_pk_mul_f32 + v_pk_fma_f32: 2 + 2 instructions per sample
// (round 3: 10)
__device__ __forceinline__ float unpack_cs8_at(uint32_t w_flipped /* word ^ 0x80808080 (the callers' form; the flip folds away) */, int k) {
    const uint32_t w = w_flipped ^ 0x80808080u;
    const float f = (float)(int8_t)(uint8_t)((w >> (8 * k)) & 0xffu);        // (b as i8) as f32, exact
    return div_small(f, kInv127Hi, kInv127Lo);
}
__device__ __forceinline__ float unpack_cu8_at(uint32_t w, int k) {
    const float f = (float)((w >> (8 * k)) & 0xffu);
    return div_small(f, kInv255Hi, kInv255Lo) - 127.5f;
}
__device__ __forceinline__ float unpack_cs8(uint32_t b) { return (float)(int8_t)(uint8_t)b / 127.0f; }
__device__ __forceinline__ float unpack_cu8(uint32_t b) {
    float q = (float)(uint8_t)b / 255.0f;
    return q - 127.5f;
}
__device__ __forceinline__ float unpack_cs16(uint32_t h) {
    const float q = div_small((float)(int16_t)(uint16_t)h, kInv65535Hi, kInv65535Lo);
    return q - 32767.5f;
}

// ---------------------------------------------------------------- glyph (src/fft.rs:45,54-60)

// `distinction` = (mx - mn) / 7.0f (src/fft.rs:45) is computed once on the host (same correctly rounded f32 division)
// and arrives as a kernel argument: left to hipcc it is hoisted out of the tile loop into a VGPR that gets spilled, and
// the reload in the epilogue carries an s_waitcnt vmcnt(0) — a full drain of the next tile's prefetch.
// The reference's arithmetic, literally: one IEEE f32 division per cell.
__device__ __forceinline__ uint8_t glyph_code_ieee(float norm, float mn, float mx, float distinction) {
    if (norm < mn) return 0;
    if (norm >= mx) return 8;
    float f = (norm - mn) / distinction;
    // Rust `as usize`: saturating, NaN -> 0
    if (!(f > 0.0f)) return 1;
    if (f >= 7.0f) return 255;   // graph[7]: the reference panics here
    return (uint8_t)(1 + (uint32_t)f);
}
// The same VALUE without the division sequence (ten instructions of the ~24 a cell costs; the glyph sink is the reference's actual
// product and its cells are as many as the norms).  q = x * RN(1/d) is the quotient to 1.2e-7 relative (half an ulp in the reciprocal, one
// rounding of the product), the IEEE quotient f = RN(x / d) to 6e-8: unless q lies within 4e-7 q of an integer, floor(q) == floor(f), f > 0
// exactly when the cell is past graph[0], f >= 7 exactly when floor(q) >= 7 — and the cell is 1 + floor(q).  Everything else (q next
// to an integer: ~1e-6 of cells; NaN; a degenerate range) takes the literal form.  tests/test_gpu_parity.py::test_glyph_equals_reference_division
// runs 2e8 norms, every threshold's neighbourhood included, through the kernels against the f32 formula.
__device__ __forceinline__ uint8_t glyph_code(float norm, float mn, float mx, float distinction, float rdist /* RN(1 / distinction), from the host like distinction */) {
    const float x = norm - mn;
    const float q = x * rdist;
    const float k = __builtin_floorf(q), fr = q - k, eps = __builtin_fmaf(q, 4e-7f, 1e-30f);
    if (__builtin_expect(!(fr > eps && fr < 1.0f - eps), 0)) return glyph_code_ieee(norm, mn, mx, distinction);
    if (norm < mn) return 0;
    if (norm >= mx) return 8;
    return k >= 7.0f ? (uint8_t)255 : (uint8_t)(1u + (uint32_t)k);
}

// ---------------------------------------------------------------- FFT butterflies (rustfft 6.4.0 scalar)

__device__ __forceinline__ void bf2(float2 &l, float2 &r) {
    float2 t = cadd(l, r);
    r = csub(l, r);
    l = t;
}

// Butterfly4::perform_fft_contiguous
__device__ __forceinline__ void bf4(float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
    float2 v0 = x0, v1 = x1, v2 = x2, v3 = x3;
    bf2(v0, v2);
    bf2(v1, v3);
    v3 = rot90(v3);
    bf2(v0, v1);
    bf2(v2, v3);
    x0 = v0; x1 = v2; x2 = v1; x3 = v3;
}

// Butterfly8::perform_fft_contiguous
__device__ __forceinline__ void bf8(float2 *v, float root2) {
    float2 a0 = v[0], a1 = v[2], a2 = v[4], a3 = v[6];
    float2 b0 = v[1], b1 = v[3], b2 = v[5], b3 = v[7];
    bf4(a0, a1, a2, a3);
    bf4(b0, b1, b2, b3);
    b1 = cscale(cadd(rot90(b1), b1), root2);
    b2 = rot90(b2);
    b3 = cscale(csub(rot90(b3), b3), root2);
    bf2(a0, b0); bf2(a1, b1); bf2(a2, b2); bf2(a3, b3);
    v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
    v[4] = b0; v[5] = b1; v[6] = b2; v[7] = b3;
}

// Butterfly16::perform_fft_contiguous (one hard-coded step of split radix)
__device__ __forceinline__ void bf16(float2 *v, float2 tw1, float2 tw2, float2 tw3, float root2) {
    float2 ev[8] = { v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14] };
    float2 p0 = v[1], p1 = v[5], p2 = v[9], p3 = v[13];
    float2 q0 = v[15], q1 = v[3], q2 = v[7], q3 = v[11];
    bf8(ev, root2);
    bf4(p0, p1, p2, p3);
    bf4(q0, q1, q2, q3);
    p1 = cmul(p1, tw1); q1 = cmul(q1, cconj(tw1));
    p2 = cmul(p2, tw2); q2 = cmul(q2, cconj(tw2));
    p3 = cmul(p3, tw3); q3 = cmul(q3, cconj(tw3));
    bf2(p0, q0); bf2(p1, q1); bf2(p2, q2); bf2(p3, q3);
    q0 = rot90(q0); q1 = rot90(q1); q2 = rot90(q2); q3 = rot90(q3);
    v[0] = cadd(ev[0], p0); v[1] = cadd(ev[1], p1); v[2] = cadd(ev[2], p2); v[3] = cadd(ev[3], p3);
    v[4] = cadd(ev[4], q0); v[5] = cadd(ev[5], q1); v[6] = cadd(ev[6], q2); v[7] = cadd(ev[7], q3);
    v[8] = csub(ev[0], p0); v[9] = csub(ev[1], p1); v[10] = csub(ev[2], p2); v[11] = csub(ev[3], p3);
    v[12] = csub(ev[4], q0); v[13] = csub(ev[5], q1); v[14] = csub(ev[6], q2); v[15] = csub(ev[7], q3);
}

// reverse the `digits` base-4 digits of x
__device__ __forceinline__ uint32_t rev4(uint32_t x, uint32_t digits) {
    uint32_t r = 0;
    for (uint32_t d = 0; d < digits; ++d) { r = (r << 2) | (x & 3u); x >>= 2; }
    return r;
}

// ---------------------------------------------------------------- NCO (src/shift.rs:49-50)
//
// Reference: place = fl((n as f64) * ratio); mul = (cos(place) as f32, sin(place) as f32).
// A libm-grade f64 sincos per sample costs more f64 issue slots than the chip has at HBM rate, and f64 work also
// lowers the clock the chip holds.  Scheme: the EXACT product X = n*ratio splits as n_r*ratio + j*ratio (n = n_r + j,
// n_r the first sample of a row of ROW samples).  Tables hold cos/sin of those two exact products (each the libm
// sincos of the rounded product, corrected to first order by the product's exactly known rounding residual), so one
// rotation gives cos/sin(X).  The reference's argument is place = fl(X) = X - r with r = fma(n, ratio, -place) the exact
// residual of ITS multiplication (|r| <= ulp(place)/2), so cos(place) = cos(X - r) = C + r*S, sin(place) = S - r*C to
// first order (the dropped r^2/2 is <= 1.1e-16 for |place| < 2^28 rad; beyond that the second-order form is used).
// 9 f64 operations per sample (12 second-order) + two f64->f32 converts; absolute error ~4e-16, so the f32 rounding
// equals glibc's except when the f64 value lies within ~1e-8 f32-ulp of a rounding boundary.

// Row-base entry: everything that is uniform over one row of ROW consecutive samples.
struct __attribute__((aligned(32))) RowBase {
    double c, s;     // cos/sin of the exact product (row*ROW) * ratio
    double nf;       // (double)(row*ROW)
    double pad_;
};

// Per-lane constants for one sample slot j inside a row: cos/sin of the exact product j * ratio.
struct LaneRot { double jf, c, s; };

// table entry: cos/sin of the exact product k * ratio, from the platform sincos of the rounded product
__device__ __forceinline__ void nco_table_entry(double kf, double ratio, double *c, double *s) {
    const double th = kf * ratio;
    const double lo = __builtin_fma(kf, ratio, -th);     // exact: k*ratio = th + lo
    double s0, c0;
    sincos(th, &s0, &c0);
    // cos / sin(th + lo), |lo| <= ulp(th)/2 (up to ~4e-6 at the far end of a 2^34-sample stream): third order in lo, the
    // small terms gathered first so that each entry takes one final rounding
    const double q = 0.5 * lo * lo, sl = __builtin_fma(-lo * lo * lo, 1.0 / 6.0, lo);     // 1 - cos lo, sin lo
    *c = c0 + __builtin_fma(-q, c0, -sl * s0);
    *s = s0 + __builtin_fma(-q, s0, sl * c0);
}

template <bool SECOND_ORDER>
__device__ __forceinline__ float2 nco_mul(const RowBase &rb, const LaneRot &lr, double ratio) {
    const double nf = rb.nf + lr.jf;                     // exact (integers < 2^53)
    const double place = nf * ratio;                     // == reference `place`
    const double r = __builtin_fma(nf, ratio, -place);   // exact: n*ratio = place + r
    const double C = __builtin_fma(-rb.s, lr.s, rb.c * lr.c);      // cos / sin of the exact product
    const double S = __builtin_fma(rb.s, lr.c, rb.c * lr.s);
    double c, s;
    if constexpr (SECOND_ORDER) {
        const double h = 0.5 * r;
        c = __builtin_fma(r, __builtin_fma(-h, C, S), C);          // C (1 - r^2/2) + r S
        s = __builtin_fma(-r, __builtin_fma(h, S, C), S);          // S (1 - r^2/2) - r C
    } else {
        c = __builtin_fma(r, S, C);
        s = __builtin_fma(-r, C, S);
    }
    return make_float2((float)c, (float)s);
}

// The same arithmetic for the N samples a lane owns, written step-by-step across the samples so
// that the N independent f64 dependency chains are issued interleaved.
template <bool SECOND_ORDER, int N>
__device__ __forceinline__ void nco_mul_n(const RowBase &rb, const LaneRot *lr, double ratio, float2 *m) {
    double nf[N], pl[N], r[N], C[N], S[N], c[N], s[N];
#pragma unroll
    for (int u = 0; u < N; ++u) nf[u] = rb.nf + lr[u].jf;
#pragma unroll
    for (int u = 0; u < N; ++u) { pl[u] = nf[u] * ratio; C[u] = rb.c * lr[u].c; S[u] = rb.c * lr[u].s; }
#pragma unroll
    for (int u = 0; u < N; ++u) { r[u] = __builtin_fma(nf[u], ratio, -pl[u]); C[u] = __builtin_fma(-rb.s, lr[u].s, C[u]); S[u] = __builtin_fma(rb.s, lr[u].c, S[u]); }
    if constexpr (SECOND_ORDER) {
        double h[N], uu[N], vv[N];
#pragma unroll
        for (int u = 0; u < N; ++u) h[u] = 0.5 * r[u];
#pragma unroll
        for (int u = 0; u < N; ++u) { uu[u] = __builtin_fma(-h[u], C[u], S[u]); vv[u] = __builtin_fma(h[u], S[u], C[u]); }
#pragma unroll
        for (int u = 0; u < N; ++u) { c[u] = __builtin_fma(r[u], uu[u], C[u]); s[u] = __builtin_fma(-r[u], vv[u], S[u]); }
    } else {
#pragma unroll
        for (int u = 0; u < N; ++u) { c[u] = __builtin_fma(r[u], S[u], C[u]); s[u] = __builtin_fma(-r[u], C[u], S[u]); }
    }
#pragma unroll
    for (int u = 0; u < N; ++u) m[u] = make_float2((float)c[u], (float)s[u]);
}

}  // namespace qd
