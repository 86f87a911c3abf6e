/*
 * quadrs_oracle.c — CPU restatement of the quadrs hot path.  TEST INFRASTRUCTURE ONLY.
 * See quadrs_oracle.h for the pinning status.  Every function cites the reference
 * file:line (paths relative to /root/reference) it follows.
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off  (no FMA contraction: Rust never
 * contracts a*b+c, and the reference's results depend on separately rounded mul/add).
 */
#include "quadrs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/lib.rs:12,23 — const TAU: f64 = PI * 2. */
static const double QO_PI64 = 3.14159265358979323846264338327950288;
#define QO_TAU64 (QO_PI64 * 2.0)
/* std::f32::consts::{PI,TAU} */
static const float QO_PI32 = 3.14159265358979323846264338327950288f;
static const float QO_TAU32 = 6.28318530717958647692528676655900577f;

static int g_closed_form = 1;
void qo_set_lowpass_closed_form(int on) { g_closed_form = on; }

/* ------------------------------------------------------------------ A1: unpack */

/* src/lib.rs:217-229 */
uint64_t qo_pair_bytes(int fmt) {
    switch (fmt) {
    case QO_FMT_CF32: return 8;
    case QO_FMT_CS8:
    case QO_FMT_CU8:  return 2;
    case QO_FMT_CS16: return 4;
    }
    return 0;
}

/* src/lib.rs:241-255 — one IEEE f32 division, then (cu8/cs16) one IEEE f32 subtraction */
float qo_to_f32(int fmt, const uint8_t *b) {
    switch (fmt) {
    case QO_FMT_CF32: {            /* LittleEndian::read_f32, :248 */
        uint32_t u = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    case QO_FMT_CS8:               /* f32::from(buf[0] as i8) / 127.0, :251 */
        return (float)(int8_t)b[0] / 127.0f;
    case QO_FMT_CU8: {             /* f32::from(buf[0]) / 255.0 - (255.0 / 2.0), :252 */
        volatile float q = (float)b[0] / 255.0f;
        return q - (255.0f / 2.0f);
    }
    case QO_FMT_CS16: {            /* f32::from(read_i16) / 65535.0 - (65535.0 / 2.0), :253 */
        int16_t v = (int16_t)((uint16_t)b[0] | ((uint16_t)b[1] << 8));
        volatile float q = (float)v / 65535.0f;
        return q - (65535.0f / 2.0f);
    }
    }
    return 0.0f;
}

/* src/lib.rs:231-238 + the loop at src/samples.rs:85-90 */
void qo_unpack(int fmt, const uint8_t *bytes, size_t n_pairs, qo_c32 *out) {
    size_t tb = (size_t)qo_pair_bytes(fmt) / 2;
    for (size_t i = 0; i < n_pairs; i++) {
        out[i].re = qo_to_f32(fmt, bytes + 2 * tb * i);
        out[i].im = qo_to_f32(fmt, bytes + 2 * tb * i + tb);
    }
}

/* ------------------------------------------------------------------ complex helpers */

/* num-complex 0.4.6 Mul: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re), no FMA */
static inline qo_c32 c_mul(qo_c32 a, qo_c32 b) {
    qo_c32 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline qo_c32 c_add(qo_c32 a, qo_c32 b) { qo_c32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline qo_c32 c_sub(qo_c32 a, qo_c32 b) { qo_c32 r = { a.re - b.re, a.im - b.im }; return r; }
static inline qo_c32 c_scale(qo_c32 a, float s) { qo_c32 r = { a.re * s, a.im * s }; return r; }

/* num-complex norm() = re.hypot(im) -> libm hypotf (src/fft.rs:53, src/ffts.rs:77) */
float qo_norm(qo_c32 v) { return hypotf(v.re, v.im); }
/* the same over an array (test helper: the device's |X| against the libm the reference links, src/fft.rs:53) */
void qo_norm_batch(const qo_c32 *v, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = hypotf(v[i].re, v[i].im); }

/*
 * Model of the DEVICE's |X| (quadrs_amd/csrc/qd_device.h, norm_ref) — not reference code: a CPU restatement of the engine's own
 * short form, so that its claim "equal to glibc hypotf for every input" is checked where 10^9 pairs cost seconds.
 * glibc 2.35 __hypotf is (float)sqrt((double)x*x + (double)y*y) (inf / nan screened first).  The device computes s the same way,
 * then, instead of an IEEE f64 sqrt (~20 f64-rate instructions), takes r = sf * rsq(sf) with sf = (float)s (about 22 good bits),
 * ONE Newton step in f64, g = r + (s - r*r) * (0.5 * rsq), whose distance from sqrt(s) is bounded by 2^-43.4 sqrt(s), and returns
 * (float)g UNLESS g lies within 2^12 f64-ulps of an f32 rounding boundary (the low 29 mantissa bits near 0x10000000) or s is outside
 * [2^-96, 2^96): then the IEEE form runs.  Away from a boundary g, sqrt(s) and RN64(sqrt(s)) round to the same f32, so the result
 * equals the reference's including ITS double rounding.  `q_ulps` perturbs the model's reciprocal square root by that many f32
 * ulps: the hardware instruction (v_rsq_f32) is accurate to 1 ulp, and the proof must hold for any such value.
 * returns the model's result; *slow = 1 when the IEEE path was taken.
 */
float qo_device_norm_model(float x, float y, int q_ulps, int *slow) {
    union { double d; uint64_t u; } S, G;
    const double dx = (double)x, dy = (double)y;
    S.d = dx * dx + dy * dy;
    if (slow) *slow = 0;
    const uint32_t hi = (uint32_t)(S.u >> 32);
    const int in_range = (uint32_t)(hi - ((1023u - 96u) << 20)) < (192u << 20);
    if (in_range) {
        const float sf = (float)S.d;
        union { float f; uint32_t u; } Q;
        Q.f = (float)(1.0 / sqrt((double)sf));
        Q.u += (uint32_t)q_ulps;                               /* any value within the instruction's accuracy */
        const float r = sf * Q.f, qh = 0.5f * Q.f;
        const double rd = (double)r;
        const double e = fma(-rd, rd, S.d);
        G.d = fma(e, (double)qh, rd);
        const uint32_t m = ((uint32_t)G.u & 0x1fffffffu) - (0x10000000u - 4096u);
        if (m >= 8192u) return (float)G.d;
    }
    if (slow) *slow = 1;
    if (!(isfinite(x) && isfinite(y))) {
        if (isinf(x) || isinf(y)) return INFINITY;
        return x + y;
    }
    return (float)sqrt(S.d);
}

/* n pseudo-random pairs (splitmix64; exponents spread over `spread` binades around 1.0, mode 1: both components of similar
 * size, mode 0: independent) through the model and through hypotf: mismatches and slow-path count */
static uint64_t qo_sm64(uint64_t *st) { uint64_t z = (*st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
uint64_t qo_device_norm_selftest(uint64_t n, uint64_t seed, int spread, int mode, int q_ulps, uint64_t *n_slow) {
    uint64_t st = seed, bad = 0, slow_n = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t h = qo_sm64(&st), h2 = qo_sm64(&st);
        union { float f; uint32_t u; } X, Y, A, B;
        const int ex = 127 + (int)((h >> 40) % (uint64_t)(2 * spread + 1)) - spread;
        const int ey = mode ? ex + (int)((h2 >> 40) % 5u) - 2 : 127 + (int)((h2 >> 40) % (uint64_t)(2 * spread + 1)) - spread;
        X.u = ((uint32_t)(h >> 63) << 31) | ((uint32_t)(ex < 1 ? 1 : (ex > 254 ? 254 : ex)) << 23) | ((uint32_t)h & 0x7fffffu);
        Y.u = ((uint32_t)(h2 >> 63) << 31) | ((uint32_t)(ey < 1 ? 1 : (ey > 254 ? 254 : ey)) << 23) | ((uint32_t)h2 & 0x7fffffu);
        int sl = 0;
        A.f = qo_device_norm_model(X.f, Y.f, q_ulps, &sl);
        B.f = hypotf(X.f, Y.f);
        slow_n += (uint64_t)sl;
        if (A.u != B.u) bad++;
    }
    if (n_slow) *n_slow = slow_n;
    return bad;
}

/* ------------------------------------------------------------------ A3: shift */

/* src/shift.rs:28 — TAU * (frequency as f64) / (sample_rate as f64) */
double qo_shift_ratio(int64_t frequency, uint64_t sample_rate) {
    return QO_TAU64 * (double)frequency / (double)sample_rate;
}

/* src/shift.rs:49-50 */
void qo_shift_multiplier(double ratio, uint64_t n, float *c, float *s) {
    double place = (double)n * ratio;
    *c = (float)cos(place);
    *s = (float)sin(place);
}

/* src/shift.rs:48-52 */
void qo_shift_apply(qo_c32 *buf, size_t n, uint64_t abs_off, double ratio) {
    for (size_t i = 0; i < n; i++) {
        qo_c32 m;
        qo_shift_multiplier(ratio, abs_off + (uint64_t)i, &m.re, &m.im);
        buf[i] = c_mul(buf[i], m);
    }
}

/* ------------------------------------------------------------------ A4: taps */

/* src/filter.rs:126-128 then `cutoff as f32` at :31 */
float qo_cutoff(uint64_t frequency, uint64_t sample_rate) {
    double c = (double)frequency / (double)sample_rate;
    return (float)c;
}

/* src/filter.rs:87-89 */
static float sinc32(float x) {
    float xp = x * QO_PI32;
    return sinf(xp) / xp;
}

/* src/filter.rs:86-105 — all f32, sequential sum, divide by sum */
void qo_lowpass_taps(float cutoff, size_t size, float *out) {
    float sz1 = (float)size - 1.0f;
    for (size_t i = 0; i < size; i++) {
        float fi = (float)i;
        /* :92-93  0.42 - 0.5*cos(2.0*PI*i/(size-1)) + 0.08*cos(4.0*PI*i/(size-1)) */
        float a1 = (2.0f * QO_PI32) * fi / sz1;
        float a2 = (4.0f * QO_PI32) * fi / sz1;
        float window = 0.42f - 0.5f * cosf(a1) + 0.08f * cosf(a2);
        /* :97  sinc(2.0 * cutoff * (i - (size-1)/2.0)) */
        float wave = sinc32(2.0f * cutoff * (fi - sz1 / 2.0f));
        out[i] = wave * window;
    }
    float sum = 0.0f;                       /* :103 */
    for (size_t i = 0; i < size; i++) sum += out[i];
    for (size_t i = 0; i < size; i++) out[i] = out[i] / sum;   /* :104 */
}

/* ------------------------------------------------------------------ A5: FIR */

/* src/filter.rs:107-124, literal.  out must hold valid + T/2 - 1 entries (when valid>=1). */
size_t qo_complex_convolve(const float *filter, size_t T, const qo_c32 *in, size_t valid, qo_c32 *out) {
    long h_len = (long)(T / 2);
    long n_in = (long)valid;
    size_t count = 0;
    for (long i = -(long)(T / 2); i < n_in - 1; i++) {
        long oi = i + h_len;
        qo_c32 acc = { 0.0f, 0.0f };
        for (long j = 0; j < (long)T; j++) {
            long ii = i + j;
            if (ii < 0 || ii >= n_in) continue;
            qo_c32 p = c_scale(in[ii], filter[j]);   /* Complex<f32> * f32 */
            acc = c_add(acc, p);                     /* += */
        }
        out[oi] = acc;
        count = (size_t)oi + 1;
    }
    return count;
}

/* src/filter.rs:68-83 on an already fetched raw block.  Returns produced count or QO_PANIC. */
size_t qo_lowpass_block(const float *taps, size_t T, uint64_t D, const qo_c32 *raw, size_t valid,
                        qo_c32 *out, size_t out_cap) {
    if (T < 2 || D == 0) return QO_PANIC;
    if (valid < T) return QO_PANIC;               /* usize underflow at :76 */
    size_t out_n = (size_t)((uint64_t)(valid - T) / D);
    if (out_n > out_cap) return QO_PANIC;         /* buf[i] out of bounds */
    if (!g_closed_form) {
        size_t conv_len = valid + T / 2 - 1;
        qo_c32 *conv = (qo_c32 *)malloc(sizeof(qo_c32) * (conv_len ? conv_len : 1));
        size_t got = qo_complex_convolve(taps, T, raw, valid, conv);
        if (got != conv_len) { free(conv); return QO_PANIC; }   /* assert_eq at :74 */
        for (size_t k = 0; k < out_n; k++) out[k] = conv[T + k * (size_t)D];   /* :78-80 */
        free(conv);
        return out_n;
    }
    /* closed form: conv[T + kD] = sum_{j<jmax} raw[kD + c + j] * h[j], c = T - T/2,
     * jmax = min(T, valid - (kD + c)); same products in the same ascending-j order. */
    size_t c = T - T / 2;
    for (size_t k = 0; k < out_n; k++) {
        size_t base = k * (size_t)D + c;
        size_t jmax = T;
        if (valid - base < jmax) jmax = valid - base;
        qo_c32 acc = { 0.0f, 0.0f };
        for (size_t j = 0; j < jmax; j++) acc = c_add(acc, c_scale(raw[base + j], taps[j]));
        out[k] = acc;
    }
    return out_n;
}

/* ------------------------------------------------------------------ A10: Samples chain */

enum { K_MEM = 1, K_GEN, K_SHIFT, K_LOWPASS };

struct qo_node {
    int kind;
    qo_node *inner;
    uint64_t sample_rate;
    /* mem */
    const uint8_t *bytes; uint64_t n_bytes; int fmt;
    /* gen */
    int64_t *cos_hz; size_t n_cos; double seconds;
    /* shift */
    double ratio;
    /* lowpass */
    float *taps; size_t T; uint64_t D;
};

qo_node *qo_source_mem(const uint8_t *bytes, uint64_t n_bytes, int fmt, uint64_t sample_rate) {
    qo_node *n = (qo_node *)calloc(1, sizeof(*n));
    n->kind = K_MEM; n->bytes = bytes; n->n_bytes = n_bytes; n->fmt = fmt; n->sample_rate = sample_rate;
    return n;
}

/* src/gen.rs:17-27 */
qo_node *qo_source_gen(const int64_t *cos_hz, size_t n_cos, uint64_t sample_rate, double seconds) {
    if (n_cos == 0 || sample_rate == 0 || !(seconds > 0.0)) return NULL;
    qo_node *n = (qo_node *)calloc(1, sizeof(*n));
    n->kind = K_GEN; n->sample_rate = sample_rate; n->seconds = seconds; n->n_cos = n_cos;
    n->cos_hz = (int64_t *)malloc(sizeof(int64_t) * n_cos);
    memcpy(n->cos_hz, cos_hz, sizeof(int64_t) * n_cos);
    return n;
}

/* src/shift.rs:19-31; Operation::exec passes orig.sample_rate() (src/lib.rs:104-105) */
qo_node *qo_shift(qo_node *inner, int64_t frequency) {
    uint64_t sr = qo_sample_rate(inner);
    int64_t af = frequency < 0 ? -frequency : frequency;
    if (!(af < (int64_t)(sr / 2))) return NULL;      /* assert at :20-23 */
    if (sr == 0) return NULL;                        /* :24 */
    qo_node *n = (qo_node *)calloc(1, sizeof(*n));
    n->kind = K_SHIFT; n->inner = inner; n->sample_rate = sr;
    n->ratio = qo_shift_ratio(frequency, sr);
    return n;
}

/* src/filter.rs:22-38; Operation::exec passes orig.sample_rate() (src/lib.rs:113-119) */
qo_node *qo_lowpass(qo_node *inner, uint64_t frequency, uint64_t decimate, size_t size) {
    qo_node *n = (qo_node *)calloc(1, sizeof(*n));
    n->kind = K_LOWPASS; n->inner = inner; n->sample_rate = qo_sample_rate(inner);
    n->T = size; n->D = decimate;
    n->taps = (float *)malloc(sizeof(float) * (size ? size : 1));
    qo_lowpass_taps(qo_cutoff(frequency, n->sample_rate), size, n->taps);
    return n;
}

void qo_free(qo_node *n) {
    while (n) {
        qo_node *in = n->inner;
        free(n->cos_hz); free(n->taps); free(n);
        n = in;
    }
}

/* Rust `x as u64` for f64: saturating, NaN -> 0 */
static uint64_t f64_as_u64(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}

uint64_t qo_len(const qo_node *n) {
    switch (n->kind) {
    case K_MEM:  return n->n_bytes / qo_pair_bytes(n->fmt);               /* src/samples.rs:64-66 */
    case K_GEN:  return f64_as_u64(n->seconds * (double)n->sample_rate);   /* src/gen.rs:31-33 */
    case K_SHIFT: return qo_len(n->inner);                                 /* src/shift.rs:38-40 */
    case K_LOWPASS: {                                                      /* src/filter.rs:45-48 */
        uint64_t il = qo_len(n->inner);
        if (il == UINT64_MAX || il < (uint64_t)n->T || n->D == 0) return UINT64_MAX;
        return 1 + (il - (uint64_t)n->T) / n->D;
    }
    }
    return UINT64_MAX;
}

uint64_t qo_sample_rate(const qo_node *n) {
    if (n->kind == K_LOWPASS) return n->D ? n->sample_rate / n->D : 0;   /* src/filter.rs:50-52 */
    return n->sample_rate;
}

size_t qo_read_at(const qo_node *n, uint64_t off, qo_c32 *buf, size_t len) {
    switch (n->kind) {
    case K_MEM: {                                   /* src/samples.rs:72-93 */
        uint64_t pb = qo_pair_bytes(n->fmt);
        if (!(off < n->n_bytes / pb)) return QO_PANIC;         /* assert at :74 */
        uint64_t wanted = pb * (uint64_t)len;
        uint64_t boff = off * pb;
        uint64_t avail = n->n_bytes - boff;         /* pread returns what is there */
        uint64_t bytes = wanted < avail ? wanted : avail;
        bytes -= bytes % pb;                        /* :84 */
        qo_unpack(n->fmt, n->bytes + boff, (size_t)(bytes / pb), buf);
        return (size_t)(bytes / pb);
    }
    case K_GEN: {                                   /* src/gen.rs:35-47 */
        for (size_t i = 0; i < len; i++) {
            double base = (double)(off + (uint64_t)i) * QO_TAU64 / (double)n->sample_rate;   /* :37 */
            qo_c32 val = { 0.0f, 0.0f };
            for (size_t k = 0; k < n->n_cos; k++) {
                double f = (double)n->cos_hz[k] * base;                 /* :40 */
                qo_c32 t = { (float)cos(f), (float)sin(f) };            /* :41 */
                val = c_add(val, t);
            }
            buf[i] = val;
        }
        return len;
    }
    case K_SHIFT: {                                 /* src/shift.rs:46-54 */
        size_t valid = qo_read_at(n->inner, off, buf, len);
        if (valid == QO_PANIC) return QO_PANIC;
        qo_shift_apply(buf, valid, off, n->ratio);
        return valid;
    }
    case K_LOWPASS: {                               /* src/filter.rs:54-83 */
        size_t underlying = len * (size_t)n->D + n->T;                  /* :68 */
        qo_c32 *raw = (qo_c32 *)calloc(underlying ? underlying : 1, sizeof(qo_c32));
        size_t valid = qo_read_at(n->inner, off * n->D, raw, underlying);   /* :71 */
        size_t r = QO_PANIC;
        if (valid != QO_PANIC) r = qo_lowpass_block(n->taps, n->T, n->D, raw, valid, buf, len);
        free(raw);
        return r;
    }
    }
    return QO_PANIC;
}

/* src/samples.rs:17-27 */
int qo_read_exact_at(const qo_node *n, uint64_t off, qo_c32 *buf, size_t len) {
    size_t got = qo_read_at(n, off, buf, len);
    if (got == QO_PANIC) return 2;
    return got == len ? 0 : 1;
}

/* ------------------------------------------------------------------ FFT (rustfft 6.4.0 Radix4 restated) */

struct qo_fft {
    size_t len, base_len;
    unsigned layers;
    qo_c32 *twiddles; size_t n_tw;
    qo_c32 tw16[3];     /* Butterfly16: compute_twiddle(1..3, 16) */
    float root2;        /* Butterfly8: (0.5f64).sqrt() as f32 */
    qo_c32 *scratch;
};

/* rustfft twiddles::compute_twiddle (forward): angle = (-2*PI/len) * index in f64; cast to f32 */
static qo_c32 compute_twiddle(size_t index, size_t fft_len) {
    double constant = -2.0 * QO_PI64 / (double)fft_len;
    double angle = constant * (double)index;
    qo_c32 r = { (float)cos(angle), (float)sin(angle) };
    return r;
}

/* rustfft twiddles::rotate_90, Forward: (im, -re) */
static inline qo_c32 rot90(qo_c32 v) { qo_c32 r = { v.im, -v.re }; return r; }

/* Butterfly2::perform_fft_strided */
static inline void bf2(qo_c32 *l, qo_c32 *r) {
    qo_c32 t = c_add(*l, *r);
    *r = c_sub(*l, *r);
    *l = t;
}

/* Butterfly4::perform_fft_contiguous */
static void bf4(qo_c32 *v) {
    qo_c32 v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    bf2(&v0, &v2);
    bf2(&v1, &v3);
    v3 = rot90(v3);
    bf2(&v0, &v1);
    bf2(&v2, &v3);
    v[0] = v0; v[1] = v2; v[2] = v1; v[3] = v3;
}

/* Butterfly8::perform_fft_contiguous */
static void bf8(qo_c32 *v, float root2) {
    qo_c32 s0[4] = { v[0], v[2], v[4], v[6] };
    qo_c32 s1[4] = { v[1], v[3], v[5], v[7] };
    bf4(s0);
    bf4(s1);
    s1[1] = c_scale(c_add(rot90(s1[1]), s1[1]), root2);
    s1[2] = rot90(s1[2]);
    s1[3] = c_scale(c_sub(rot90(s1[3]), s1[3]), root2);
    for (int i = 0; i < 4; i++) bf2(&s0[i], &s1[i]);
    for (int i = 0; i < 4; i++) { v[i] = s0[i]; v[i + 4] = s1[i]; }
}

static inline qo_c32 c_conj(qo_c32 a) { qo_c32 r = { a.re, -a.im }; return r; }

/* Butterfly16::perform_fft_contiguous (one hard-coded step of split radix) */
static void bf16(qo_c32 *v, const qo_c32 *tw, float root2) {
    qo_c32 ev[8] = { v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14] };
    qo_c32 n1[4] = { v[1], v[5], v[9], v[13] };
    qo_c32 n3[4] = { v[15], v[3], v[7], v[11] };
    bf8(ev, root2);
    bf4(n1);
    bf4(n3);
    n1[1] = c_mul(n1[1], tw[0]);  n3[1] = c_mul(n3[1], c_conj(tw[0]));
    n1[2] = c_mul(n1[2], tw[1]);  n3[2] = c_mul(n3[2], c_conj(tw[1]));
    n1[3] = c_mul(n1[3], tw[2]);  n3[3] = c_mul(n3[3], c_conj(tw[2]));
    for (int i = 0; i < 4; i++) bf2(&n1[i], &n3[i]);
    for (int i = 0; i < 4; i++) n3[i] = rot90(n3[i]);
    for (int i = 0; i < 4; i++) {
        v[i]      = c_add(ev[i], n1[i]);
        v[i + 4]  = c_add(ev[i + 4], n3[i]);
        v[i + 8]  = c_sub(ev[i], n1[i]);
        v[i + 12] = c_sub(ev[i + 4], n3[i]);
    }
}

/* Radix4::new: base by exponent: 0->1, 1->2, 2->4, 3->8, else odd->8, even->16 */
qo_fft *qo_fft_new(size_t len) {
    if (len == 0 || (len & (len - 1))) return NULL;
    unsigned e = 0;
    while (((size_t)1 << e) < len) e++;
    unsigned be = e <= 3 ? e : ((e & 1) ? 3 : 4);
    qo_fft *p = (qo_fft *)calloc(1, sizeof(*p));
    p->len = len; p->base_len = (size_t)1 << be; p->layers = (e - be) / 2;
    p->root2 = (float)sqrt(0.5);
    for (int k = 0; k < 3; k++) p->tw16[k] = compute_twiddle((size_t)k + 1, 16);
    /* cross-layer twiddles, bottom layer first: for i<num_columns, for k in 1..4 */
    p->twiddles = (qo_c32 *)malloc(sizeof(qo_c32) * (len ? len : 1));
    size_t cross = p->base_len, n = 0;
    while (cross < len) {
        size_t cols = cross;
        cross *= 4;
        for (size_t i = 0; i < cols; i++)
            for (size_t k = 1; k < 4; k++) p->twiddles[n++] = compute_twiddle(i * k, cross);
    }
    p->n_tw = n;
    p->scratch = (qo_c32 *)malloc(sizeof(qo_c32) * len);
    return p;
}

void qo_fft_free(qo_fft *p) { if (p) { free(p->twiddles); free(p->scratch); free(p); } }
size_t qo_fft_twiddle_count(const qo_fft *p) { return p->n_tw; }
const qo_c32 *qo_fft_twiddles(const qo_fft *p) { return p->twiddles; }
size_t qo_fft_base_len(const qo_fft *p) { return p->base_len; }

/* reverse the base-4 digits of x (rev_digits digits) */
static size_t rev4(size_t x, unsigned digits) {
    size_t r = 0;
    for (unsigned d = 0; d < digits; d++) { r = (r << 2) | (x & 3); x >>= 2; }
    return r;
}

void qo_fft_process(const qo_fft *p, qo_c32 *buf) {
    size_t len = p->len, base = p->base_len;
    qo_c32 *out = p->scratch;
    if (len == base) {
        memcpy(out, buf, sizeof(qo_c32) * len);
    } else {
        /* bitreversed_transpose::<_, 4>(base_len, input, output):
         * output[y + rev(x)*height] = input[x + y*width], width = len/base */
        size_t width = len / base;
        for (size_t x = 0; x < width; x++) {
            size_t xr = rev4(x, p->layers);
            for (size_t y = 0; y < base; y++) out[y + xr * base] = buf[x + y * width];
        }
    }
    /* base FFTs over contiguous chunks */
    for (size_t c = 0; c < len; c += base) {
        switch (base) {
        case 1: break;
        case 2: bf2(&out[c], &out[c + 1]); break;
        case 4: bf4(&out[c]); break;
        case 8: bf8(&out[c], p->root2); break;
        case 16: bf16(&out[c], p->tw16, p->root2); break;
        }
    }
    /* radix-4 cross layers (butterfly_4: twiddle the three upper rows, then Butterfly4) */
    size_t cross = base;
    const qo_c32 *tw = p->twiddles;
    while (cross < len) {
        size_t cols = cross;
        cross *= 4;
        for (size_t chunk = 0; chunk < len; chunk += cross) {
            qo_c32 *d = out + chunk;
            for (size_t i = 0; i < cols; i++) {
                qo_c32 s[4];
                s[0] = d[i];
                s[1] = c_mul(d[i + cols], tw[3 * i]);
                s[2] = c_mul(d[i + 2 * cols], tw[3 * i + 1]);
                s[3] = c_mul(d[i + 3 * cols], tw[3 * i + 2]);
                bf4(s);
                d[i] = s[0]; d[i + cols] = s[1]; d[i + 2 * cols] = s[2]; d[i + 3 * cols] = s[3];
            }
        }
        tw += 3 * cols;
    }
    memcpy(buf, out, sizeof(qo_c32) * len);
}

void qo_dft_f64(const qo_c32 *in, size_t len, double *out_re, double *out_im) {
    /* the len roots of unity once (same values the per-term cos/sin gave; (k*n) mod len indexes them) */
    double *rc = (double *)malloc(sizeof(double) * (len ? len : 1)), *rs = (double *)malloc(sizeof(double) * (len ? len : 1));
    for (size_t m = 0; m < len; m++) {
        double a = -2.0 * QO_PI64 * (double)m / (double)len;
        rc[m] = cos(a); rs[m] = sin(a);
    }
    for (size_t k = 0; k < len; k++) {
        double sr = 0.0, si = 0.0;
        size_t m = 0;                                   /* (k*n) mod len, stepped */
        for (size_t n = 0; n < len; n++) {
            double c = rc[m], s = rs[m];
            sr += (double)in[n].re * c - (double)in[n].im * s;
            si += (double)in[n].re * s + (double)in[n].im * c;
            m += k; if (m >= len) m -= len;
        }
        out_re[k] = sr; out_im[k] = si;
    }
    free(rc); free(rs);
}

/* ------------------------------------------------------------------ A6: spark_fft */

/* Rust `x as usize` for f32: saturating, NaN -> 0 */
static size_t f32_as_usize(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)x;
}

/* src/fft.rs:45,53-60 */
uint8_t qo_glyph_code(float norm, float min, float max) {
    float distinction = (max - min) / 7.0f;        /* graph.len() as f32 */
    if (norm < min) return 0;
    if (norm >= max) return 8;
    size_t idx = f32_as_usize((norm - min) / distinction);
    if (idx >= 7) return 255;                      /* graph[7] -> index out of bounds panic */
    return (uint8_t)(1 + idx);
}

/* trip count of `i = 0; while i < len - W { ...; i += S }` (src/fft.rs:27-28,65) */
uint64_t qo_spark_window_count(uint64_t len, uint64_t W, uint64_t S) {
    if (len < W || S == 0) return UINT64_MAX;      /* u64 underflow */
    uint64_t lim = len - W;
    return lim == 0 ? 0 : (lim - 1) / S + 1;
}

uint64_t qo_spark_fft(const qo_node *n, size_t W, uint64_t S, int has_range, float min, float max,
                      uint64_t first_window, uint64_t cap_windows, float *norms, uint8_t *codes) {
    if (!has_range) { min = 0.08f; max = 1.0f; }   /* src/fft.rs:22-23 */
    uint64_t len = qo_len(n);
    if (len == UINT64_MAX) return UINT64_MAX;
    uint64_t total = qo_spark_window_count(len, W, S);
    if (total == UINT64_MAX) return UINT64_MAX;
    qo_fft *fft = qo_fft_new(W);                   /* Radix4::new(W, Forward), :25 */
    if (!fft) return UINT64_MAX;
    qo_c32 *inp = (qo_c32 *)malloc(sizeof(qo_c32) * W);
    uint64_t done = 0;
    for (uint64_t w = first_window; w < total && done < cap_windows; w++, done++) {
        uint64_t i = w * S;
        memset(inp, 0, sizeof(qo_c32) * W);
        if (qo_read_exact_at(n, i, inp, W) != 0) { done = UINT64_MAX; break; }   /* :30 */
        qo_fft_process(fft, inp);                  /* :32 */
        for (size_t b = 0; b < W; b++) {           /* fftshift order :48-51 */
            size_t src = (b + W / 2) % W;
            float nm = qo_norm(inp[src]);
            if (norms) norms[done * W + b] = nm;
            if (codes) codes[done * W + b] = qo_glyph_code(nm, min, max);
        }
    }
    free(inp);
    qo_fft_free(fft);
    return done;
}

/* src/fft.rs:19,63 — "sparkfft sample_rate={}\n" then "│{}│\n" per window */
size_t qo_spark_render(uint64_t sample_rate, const uint8_t *codes, uint64_t nwin, size_t W,
                       char *out, size_t cap) {
    static const char *bar = "\xE2\x94\x82";       /* │ U+2502 */
    size_t n = 0;
    char hdr[64];
    int hl = 0;
    {   /* decimal u64 */
        char tmp[32]; int t = 0; uint64_t v = sample_rate;
        do { tmp[t++] = (char)('0' + v % 10); v /= 10; } while (v);
        const char *pre = "sparkfft sample_rate=";
        for (const char *p = pre; *p; p++) hdr[hl++] = *p;
        while (t) hdr[hl++] = tmp[--t];
        hdr[hl++] = '\n';
    }
#define PUT(c) do { if (out && n < cap) out[n] = (char)(c); n++; } while (0)
    for (int i = 0; i < hl; i++) PUT(hdr[i]);
    for (uint64_t w = 0; w < nwin; w++) {
        PUT(bar[0]); PUT(bar[1]); PUT(bar[2]);
        for (size_t b = 0; b < W; b++) {
            uint8_t c = codes[w * W + b];
            if (c == 0) { PUT(' '); }
            else if (c <= 8) { PUT(0xE2); PUT(0x96); PUT(0x80 + c); }   /* ▁..▇ = U+2581..7, █ = U+2588 */
            else { PUT('?'); }
        }
        PUT(bar[0]); PUT(bar[1]); PUT(bar[2]);
        PUT('\n');
    }
#undef PUT
    return n;
}

/* ------------------------------------------------------------------ A7: freq_levels */

/* src/fft.rs:77-101 */
uint64_t qo_freq_levels(const qo_node *n, size_t W, uint64_t S, uint64_t cap, uint8_t *vals) {
    uint64_t len = qo_len(n);
    if (len == UINT64_MAX || len < W || S == 0) return UINT64_MAX;
    qo_fft *fft = qo_fft_new(W);
    if (!fft) return UINT64_MAX;
    uint64_t total = (len - W) / S;                /* :86 */
    qo_c32 *inp = (qo_c32 *)malloc(sizeof(qo_c32) * W);
    uint64_t done = 0;
    for (uint64_t r = 0; r < total && r < cap; r++, done++) {
        memset(inp, 0, sizeof(qo_c32) * W);
        if (qo_read_exact_at(n, r * S, inp, W) != 0) { done = UINT64_MAX; break; }  /* unwrap, :91 */
        qo_fft_process(fft, inp);
        float first = 0.0f, second = 0.0f;         /* sequential f32 sums, :95-96 */
        for (size_t k = 0; k < W / 2; k++) first += qo_norm(inp[k]);
        for (size_t k = W / 2; k < W; k++) second += qo_norm(inp[k]);
        vals[r] = first < second ? 0 : 1;          /* :97 */
    }
    free(inp);
    qo_fft_free(fft);
    return done;
}

/* ------------------------------------------------------------------ A8: take_fft */

/* src/ffts.rs:110-119 */
void qo_blackman_harris(size_t n, float *w) {
    for (size_t i = 0; i < n; i++) {
        float x = QO_TAU32 * (float)i / (float)(n - 1);
        float value = 0.35875f - 0.48829f * cosf(x) + 0.14128f * cosf(2.0f * x) - 0.01168f * cosf(3.0f * x);
        w[i] = value;
    }
}

/* src/ffts.rs:18-85.  rustfft's planner is replaced by the Radix4 restatement (power-of-two W). */
int qo_take_fft(const qo_node *n, int has_slice, uint64_t start, uint64_t end, size_t W,
                int windowing, size_t output_len, float *rows, uint64_t *row_offsets) {
    uint64_t len = qo_len(n);
    if (len == UINT64_MAX) return 2;
    if (!has_slice) { if (len < W) return 2; start = 0; end = len - (uint64_t)W; }   /* :27-30 */
    if (!(end > start)) return 2;                  /* :32-35 */
    if (!(end < len)) return 2;                    /* :36-40 */
    uint64_t visible = end - start;
    if (!(visible > (uint64_t)output_len)) return 1;   /* ensure!, :45-48 */
    /* :25 FftPlanner::plan_fft_forward(W) takes ANY length.  A power of two goes to the Radix4 restatement (what the
     * planner itself picks on a scalar target).  For every other length rustfft's planner composes mixed-radix / Rader /
     * Bluestein steps chosen by its host-SIMD build, and its source is not in /root/reference: PARITY UNPINNED.  The oracle
     * then gives the mathematically exact answer instead — an f64 DFT rounded once to f32 — and the tests compare the
     * GPU against it with an explicit error bound, not bit for bit. */
    qo_fft *fft = qo_fft_new(W);
    double *dre = NULL, *dim = NULL;
    if (!fft) {
        if (W == 0) return 2;
        dre = (double *)malloc(sizeof(double) * W); dim = (double *)malloc(sizeof(double) * W);
    }
    double step = (double)visible / (double)output_len;    /* :50 */
    qo_c32 *cb = (qo_c32 *)calloc(W, sizeof(qo_c32));
    float *win = NULL;
    if (windowing == 1) { win = (float *)malloc(sizeof(float) * W); qo_blackman_harris(W, win); }
    int rc = 0;
    for (size_t i = 0; i < output_len; i++) {
        uint64_t idx = start + f64_as_u64(round(step * (double)i));   /* :60 */
        if (row_offsets) row_offsets[i] = idx;
        int e = qo_read_exact_at(n, idx, cb, W);
        if (e) { rc = e; break; }
        if (win) for (size_t k = 0; k < W; k++) cb[k] = c_scale(cb[k], win[k]);   /* :64-68 */
        if (fft) qo_fft_process(fft, cb);
        else {
            qo_dft_f64(cb, W, dre, dim);
            for (size_t k = 0; k < W; k++) { cb[k].re = (float)dre[k]; cb[k].im = (float)dim[k]; }
        }
        for (size_t b = 0; b < W; b++) rows[i * W + b] = qo_norm(cb[(b + W / 2) % W]);   /* :72-78 */
    }
    free(cb); free(win); free(dre); free(dim);
    if (fft) qo_fft_free(fft);
    return rc;
}

/* ------------------------------------------------------------------ N1: do_write block loop */

/* src/lib.rs:199-210 */
int qo_do_write(const qo_node *n, qo_c32 *out, uint64_t cap, uint64_t *n_written) {
    uint64_t len = qo_len(n);
    *n_written = 0;
    if (len == UINT64_MAX) return 2;
    qo_c32 *buf = (qo_c32 *)malloc(sizeof(qo_c32) * 0x1000);
    uint64_t off = 0;
    int rc = 0;
    while (off < len) {
        memset(buf, 0, sizeof(qo_c32) * 0x1000);
        size_t rd = qo_read_at(n, off, buf, 0x1000);
        if (rd == QO_PANIC || rd == 0) { rc = 2; break; }   /* assert_ne!(0, read), :203 */
        off += rd;
        for (size_t i = 0; i < rd; i++) {
            if (*n_written < cap) out[*n_written] = buf[i];
            (*n_written)++;
        }
    }
    free(buf);
    return rc;
}
