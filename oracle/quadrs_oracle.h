/*
 * quadrs_oracle.h — CPU restatement of the quadrs hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X engine.  It restates, in plain C, what the
 * reference (FauxFaux/quadrs, Rust) computes on the CPU for
 *     unpack -> shift -> lowpass (FIR + decimate) -> strided short FFT -> |X| -> glyph/bucket
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under quadrs_amd/ (the product) links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - unpack / shift / taps / FIR+decimate / gen / write: restated line by line from the
 *     reference sources cited at each function; libm calls are the same glibc sin/cos/
 *     sinf/cosf/hypotf a Rust build on this image would call.  The reference has no
 *     tests or golden vectors on this path; the only externally authored known-answer is
 *     the README OOK string (README.md:113-116,167), which tests/test_oracle_golden.py
 *     reproduces exactly through this oracle.
 *   - FFT rounding order: the arithmetic lives in rustfft 6.4.0 (Cargo.lock:3207-3219),
 *     whose source is NOT in /root/reference.  qo_fft_radix4() restates rustfft's published
 *     scalar Radix4 algorithm (Butterfly1/2/4/8/16 base + radix-4 DIT cross layers,
 *     twiddles = (f32)cos/sin of an f64 angle).  No reference fixture pins its internal
 *     rounding => "parity unpinned" for FFT bit patterns; it is bounded against an f64 DFT.
 */
#ifndef QUADRS_ORACLE_H
#define QUADRS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } qo_c32;

/* FileFormat, src/lib.rs:61-74 */
enum { QO_FMT_CF32 = 0, QO_FMT_CS8 = 1, QO_FMT_CU8 = 2, QO_FMT_CS16 = 3 };

/* returned by read_at-like functions where the reference would panic */
#define QO_PANIC ((size_t)-1)

/* ---- A1: unpack (src/lib.rs:215-256) ---- */
uint64_t qo_pair_bytes(int fmt);
float    qo_to_f32(int fmt, const uint8_t *b);
void     qo_unpack(int fmt, const uint8_t *bytes, size_t n_pairs, qo_c32 *out);

/* ---- A10: the Samples chain (src/samples.rs:11-42) as a small node graph ---- */
typedef struct qo_node qo_node;

/* SampleFile over an in-memory byte image of the file (src/samples.rs:44-94) */
qo_node *qo_source_mem(const uint8_t *bytes, uint64_t n_bytes, int fmt, uint64_t sample_rate);
/* Gen (src/gen.rs:16-52); returns NULL where Gen::new would Err */
qo_node *qo_source_gen(const int64_t *cos_hz, size_t n_cos, uint64_t sample_rate, double seconds);
/* Shift::new (src/shift.rs:19-31); returns NULL where it would panic */
qo_node *qo_shift(qo_node *inner, int64_t frequency);
/* LowPass::new (src/filter.rs:22-38) */
qo_node *qo_lowpass(qo_node *inner, uint64_t frequency, uint64_t decimate, size_t size);
void     qo_free(qo_node *n);                 /* frees the whole chain */

uint64_t qo_len(const qo_node *n);            /* Samples::len; UINT64_MAX where it would panic */
uint64_t qo_sample_rate(const qo_node *n);
size_t   qo_read_at(const qo_node *n, uint64_t off, qo_c32 *buf, size_t len);
/* 0 = Ok, 1 = Err (short), 2 = panic  (src/samples.rs:17-27) */
int      qo_read_exact_at(const qo_node *n, uint64_t off, qo_c32 *buf, size_t len);

/* select LowPass::read_at implementation: 0 = literal complex_convolve over every
 * non-decimated position (src/filter.rs:107-124, the reference's cost class);
 * 1 = closed form (same products, same order, only the kept outputs). Bit-identical. */
void     qo_set_lowpass_closed_form(int on);

/* ---- A3 pieces exposed for fixtures ---- */
double   qo_shift_ratio(int64_t frequency, uint64_t sample_rate);     /* src/shift.rs:28 */
void     qo_shift_multiplier(double ratio, uint64_t n, float *c, float *s); /* src/shift.rs:49-50 */
void     qo_shift_apply(qo_c32 *buf, size_t n, uint64_t abs_off, double ratio);

/* ---- A4: taps (src/filter.rs:86-105,126-128) ---- */
float    qo_cutoff(uint64_t frequency, uint64_t sample_rate);
void     qo_lowpass_taps(float cutoff, size_t size, float *out);

/* ---- A5: complex_convolve literal (src/filter.rs:107-124); out has valid+T/2-1 entries ---- */
size_t   qo_complex_convolve(const float *filter, size_t T, const qo_c32 *in, size_t valid, qo_c32 *out);
/* LowPass::read_at on an already-fetched raw block (src/filter.rs:68-83) */
size_t   qo_lowpass_block(const float *taps, size_t T, uint64_t D, const qo_c32 *raw, size_t valid,
                          qo_c32 *out, size_t out_cap);

/* ---- FFT: restated rustfft 6.4.0 scalar Radix4 (parity unpinned, see header) ---- */
typedef struct qo_fft qo_fft;
qo_fft  *qo_fft_new(size_t len);              /* NULL unless len is a power of two */
void     qo_fft_free(qo_fft *p);
void     qo_fft_process(const qo_fft *p, qo_c32 *buf);       /* in place, forward, unnormalised */
size_t   qo_fft_twiddle_count(const qo_fft *p);
const qo_c32 *qo_fft_twiddles(const qo_fft *p);
size_t   qo_fft_base_len(const qo_fft *p);
/* f64 truth: naive forward DFT of f32 input, f64 accumulate, f64 out */
void     qo_dft_f64(const qo_c32 *in, size_t len, double *out_re, double *out_im);

/* ---- A6: spark_fft (src/fft.rs:12-69) ----
 * Writes, per window, W fftshifted norms (norms may be NULL) and W glyph codes
 * (codes may be NULL): 0=' ', 1..7='▁'..'▇', 8='█', 255 = the reference would panic
 * (index 7 into a 7-entry table, src/fft.rs:59).  cap_windows bounds the outputs;
 * first_window/stop lets a caller take a sub-range.  Returns windows produced, or
 * (uint64_t)-1 on read_exact Err / underflow (len < W). */
uint64_t qo_spark_window_count(uint64_t len, uint64_t W, uint64_t S);  /* loop trip count of :28,65 */
uint64_t qo_spark_fft(const qo_node *n, size_t W, uint64_t S, int has_range, float min, float max,
                      uint64_t first_window, uint64_t cap_windows, float *norms, uint8_t *codes);
uint8_t  qo_glyph_code(float norm, float min, float max);
/* render the exact stdout bytes of spark_fft for codes[nwin*W]: header + rows; returns bytes
 * written (call with out=NULL to size) */
size_t   qo_spark_render(uint64_t sample_rate, const uint8_t *codes, uint64_t nwin, size_t W,
                         char *out, size_t cap);

/* ---- A7: freq_levels (src/fft.rs:77-101); vals[total]; returns total or (uint64_t)-1 ---- */
uint64_t qo_freq_levels(const qo_node *n, size_t W, uint64_t S, uint64_t cap, uint8_t *vals);

/* ---- A8: take_fft (src/ffts.rs:18-85), power-of-two widths; windowing 0=Rect 1=BlackmanHarris.
 * rows[output_len*W]; returns 0 ok, 1 Err, 2 panic ---- */
void     qo_blackman_harris(size_t n, float *w);             /* src/ffts.rs:110-119 */
int      qo_take_fft(const qo_node *n, int has_slice, uint64_t start, uint64_t end, size_t W,
                     int windowing, size_t output_len, float *rows, uint64_t *row_offsets);

/* ---- N1: do_write's block loop (src/lib.rs:199-210) into memory.
 * Fills out[cap] with the cf32 samples the reference would have written before it either
 * finishes (returns 0) or hits assert_ne!(0, read) (returns 2); *n_written = samples. ---- */
int      qo_do_write(const qo_node *n, qo_c32 *out, uint64_t cap, uint64_t *n_written);

/* hypotf as the reference calls it (num-complex norm() -> f32::hypot -> libm hypotf) */
float    qo_norm(qo_c32 v);
void     qo_norm_batch(const qo_c32 *v, size_t n, float *out);
/* NOT reference code: CPU model of the device's short-form |X| (qd_device.h norm_ref) and its self-test against hypotf;
 * see quadrs_oracle.c */
float    qo_device_norm_model(float x, float y, int q_ulps, int *slow);
uint64_t qo_device_norm_selftest(uint64_t n, uint64_t seed, int spread, int mode, int q_ulps, uint64_t *n_slow);

#ifdef __cplusplus
}
#endif
#endif
