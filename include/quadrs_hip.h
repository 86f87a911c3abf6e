/*
 * quadrs_hip.h — C ABI of the MI355X (gfx950) engine for quadrs' IQ-stream hot path.
 *
 * The reference (FauxFaux/quadrs, Rust) has no FFI seam; the seam this library honours is
 * the set of constructors / sinks `Operation::exec` calls (src/lib.rs:83-175) and the
 * `Samples` sample-block iterator (src/samples.rs:11-28).  Every entry point below names
 * the reference item it replaces.  All pointers are plain C; no C++/torch types cross
 * the boundary.  Nothing here aborts or unwinds: every function returns a qd_status and
 * qd_last_error() gives a thread-local message.
 *
 * Two granularities, same kernels underneath:
 *   fine-grained  — mirrors `read_at` so shift.rs / filter.rs / fft.rs stay thin shims;
 *   coarse (plan) — one call covers a batch of FFT windows of the fused chain
 *                   unpack -> shift -> lowpass -> FFT -> |X| -> epilogue.
 *
 * Data ABI: qd_c32 == num_complex::Complex<f32> == {f32 re, f32 im}, little endian.
 */
#ifndef QUADRS_HIP_H
#define QUADRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } qd_c32;

typedef enum {
    QD_OK = 0,
    QD_ERR_INVALID = 1,      /* bad argument (anyhow::Error class in the reference) */
    QD_ERR_PANIC = 2,        /* the reference would panic here (assert!/unwrap/index) */
    QD_ERR_SHORT = 3,        /* read_exact_at Err: fewer samples than asked (src/samples.rs:17-27) */
    QD_ERR_HIP = 4,          /* a hip* call failed */
    QD_ERR_UNSUPPORTED = 5   /* valid in the reference, not built here yet */
} qd_status;

/* FileFormat, src/lib.rs:61-74 */
typedef enum { QD_FMT_CF32 = 0, QD_FMT_CS8 = 1, QD_FMT_CU8 = 2, QD_FMT_CS16 = 3 } qd_format;

/* where a buffer lives */
typedef enum { QD_MEM_HOST = 0, QD_MEM_DEVICE = 1 } qd_mem;

/* what the fused chain leaves per FFT window */
typedef enum {
    QD_EPI_NORMS_F32 = 0,    /* W f32: hypot(re,im) in fftshift order (src/fft.rs:48-53) */
    QD_EPI_GLYPH_U8 = 1,     /* W u8: 0=' ' 1..7='▁'..'▇' 8='█' 255=reference would panic (src/fft.rs:54-60) */
    QD_EPI_BUCKET2_U8 = 2,   /* 1 u8: freq_levels digit (src/fft.rs:95-97) */
    QD_EPI_CF32_BLOCKS = 3   /* the write sink (do_write, src/lib.rs:199-210): no FFT; a "window" is one full
                                LowPass::read_at block of `width` decimated samples (0x1000 for do_write, a power
                                of two), `stride` is ignored, the output is width qd_c32 per block with the block's
                                own tail truncation.  n_windows counts the FULL blocks, floor((n-T)/(width*D));
                                the ragged end of the stream is left to qd_lowpass_block.  Needs has_lowpass. */
} qd_epilogue;

const char *qd_last_error(void);
const char *qd_version(void);
int qd_device_count(int *count);
int qd_set_device(int device);

/* ------------------------------------------------------------------ fine-grained */

/* FileFormat::pair_bytes, src/lib.rs:226-229 */
uint64_t qd_pair_bytes(int fmt);

/* FileFormat::to_cf32 over a block — the loop at src/samples.rs:85-90 (bit-exact). */
int qd_unpack(int fmt, const void *bytes, size_t n_pairs, qd_c32 *out, int mem);

/* Shift::new's ratio, src/shift.rs:28 (host arithmetic, f64). */
double qd_shift_ratio(int64_t frequency, uint64_t sample_rate);

/* Shift::read_at's loop, src/shift.rs:48-52: buf[i] *= e^{+i (abs_off+i)*ratio}, in place;
 * abs_off is the absolute index of buf[0] in the stream (the NCO phase depends on it). */
int qd_shift(qd_c32 *buf, size_t n, uint64_t abs_off, double ratio, int mem);

/* lowpass_filter(cutoff_from_frequency(f, sr) as f32, size), src/filter.rs:29-31,86-105,126-128.
 * Host arithmetic with the platform libm, exactly as the reference does it (O(size), once). */
int qd_lowpass_design(uint64_t frequency, uint64_t sample_rate, size_t size, float *taps);

/* LowPass::read_at on an already fetched raw block, src/filter.rs:68-83 + complex_convolve
 * :107-124: out[k] = sum_{j<jmax(k)} raw[k*D + c + j]*taps[j], c = T - T/2,
 * jmax = min(T, valid - (k*D + c)); *produced = (valid - T)/D.  QD_ERR_PANIC if valid < T
 * or out_cap < *produced.  Same products, same ascending-j order, no FMA. */
int qd_lowpass_block(const float *taps, size_t T, uint64_t D, const qd_c32 *raw, size_t valid,
                     qd_c32 *out, size_t out_cap, size_t *produced, int mem);

/* Radix4::new(W, Forward) + process + fftshift + norm for n_fft windows, window i starting
 * at in[i*in_stride] (src/fft.rs:25,32,48-53).  norms: n_fft*W f32. */
int qd_fft_norm_batch(const qd_c32 *in, size_t W, size_t n_fft, size_t in_stride, float *norms, int mem);

/* take_fft, src/ffts.rs:18-85 (the spectrogram rows of the egui front end): output_len rows at
 * sample start + round(step*i), step = (end-start)/output_len in f64; optional Blackman-Harris
 * window (windowing 1; src/ffts.rs:110-119, host f32 arithmetic like the reference); forward FFT;
 * fftshifted norms into rows[output_len*W].  `in` holds samples [in_first, in_first+n_in) of the
 * cf32 Samples being viewed, whose len() is samples_len.  has_slice 0 => (0, len - W) (:27-30).
 * Power-of-two W only: the reference's FftPlanner also takes other lengths (not built).
 * QD_ERR_PANIC / QD_ERR_INVALID mirror the asserts / ensure! at :32-48. */
int qd_take_fft(const qd_c32 *in, uint64_t in_first, size_t n_in, uint64_t samples_len, int has_slice,
                uint64_t start, uint64_t end, size_t W, int windowing, size_t output_len, float *rows, int mem);

/* ------------------------------------------------------------------ coarse-grained plan */

typedef struct qd_plan qd_plan;

typedef struct {
    uint32_t struct_size;     /* sizeof(qd_chain_desc) */
    int32_t  format;          /* qd_format of the source bytes (Operation::From, src/lib.rs:89-96) */
    uint64_t sample_rate;     /* of the source */
    uint64_t n_samples;       /* Samples::len() of the source: the whole stream, absolute indexing */
    int32_t  has_shift;       /* Operation::Shift, src/lib.rs:102-106 */
    int32_t  _pad0;
    int64_t  shift_hz;
    int32_t  has_lowpass;     /* Operation::LowPass, src/lib.rs:107-121 */
    int32_t  _pad1;
    uint64_t lowpass_hz;
    uint64_t decimate;
    uint64_t taps;            /* `size`: 2*power, default 40 (src/args.rs:161-166) */
    uint64_t width;           /* Operation::SparkFft / Bucket, src/lib.rs:122-160 */
    uint64_t stride;
    int32_t  epilogue;        /* qd_epilogue */
    int32_t  has_range;       /* sparkfft -range min:max; else 0.08 / 1.0 (src/fft.rs:22-23) */
    float    range_min, range_max;
} qd_chain_desc;

typedef struct {
    uint64_t n_windows;       /* trip count of the sink's loop: spark_fft `while i < len - W`
                                 (src/fft.rs:28,65) or freq_levels `(len - W)/S` (src/fft.rs:86) */
    uint64_t decimated_len;   /* Samples::len() seen by the sink (LowPass::len, src/filter.rs:45-48) */
    uint64_t out_sample_rate; /* Samples::sample_rate() seen by the sink */
    uint64_t out_bytes_per_window;
    uint64_t raw_per_window;  /* W*D + T source samples one window reads */
    uint64_t raw_step;        /* S*D source samples between window starts */
    double   ratio;           /* Shift ratio (0 if no shift) */
    uint32_t tile_windows;    /* windows per workgroup tile */
    uint32_t threads;         /* workgroup size */
    uint32_t lds_bytes;
    uint32_t kernel_kind;     /* 0 generic (runtime geometry), 1 built-in shape-specialised, 2 specialised at plan time (hiprtc) */
} qd_plan_info;

int qd_plan_create(const qd_chain_desc *desc, qd_plan **plan);
int qd_plan_destroy(qd_plan *plan);
int qd_plan_get_info(const qd_plan *plan, qd_plan_info *info);
/* taps the plan designed (T floats), for inspection */
int qd_plan_get_taps(const qd_plan *plan, float *taps, size_t cap);

/* source samples [*first, *first + *count) that windows [first_window, first_window+n) read */
int qd_plan_src_range(const qd_plan *plan, uint64_t first_window, uint64_t n_windows,
                      uint64_t *first, uint64_t *count);

/* Run windows [first_window, first_window + n_windows) of the sink's loop.
 * src holds the raw bytes of source samples [src_first, src_first + src_count) (a slab of the
 * stream; absolute sample indices keep the NCO phase identical to a whole-stream run).
 * out receives n_windows * out_bytes_per_window bytes.
 * Device buffers: the kernels are enqueued on `stream` (a hipStream_t, may be NULL) and the
 * call returns without synchronising.  Host buffers: chunked, double-buffered
 * hipMemcpyAsync in and out; returns after the last copy completed. */
int qd_plan_run(qd_plan *plan, const void *src, int src_mem, uint64_t src_first, uint64_t src_count,
                uint64_t first_window, uint64_t n_windows, void *out, int out_mem, void *stream);

/* HIP-event timing of the chain kernel of the most recent device-resident qd_plan_run,
 * taken on the stream it was launched on.  Enable before the run; the query synchronises. */
int qd_plan_set_timing(qd_plan *plan, int enabled);
int qd_plan_last_kernel_ms(qd_plan *plan, float *ms);

/* Device buffers for callers that keep a stream resident in HBM — e.g. `gen ... | lowpass | sparkfft`
 * (src/gen.rs + BASELINE configs[3]): the samples are generated on the device by qd_gen and never cross PCIe.
 * Plain hipMalloc / hipFree / hipMemcpy behind the ABI so that a host without a HIP toolchain can use
 * qd_plan_run's device path.  qd_device_copy is synchronous; *_mem are qd_mem values. */
int qd_device_alloc(size_t bytes, void **ptr);
int qd_device_free(void *ptr);
int qd_device_copy(void *dst, int dst_mem, const void *src, int src_mem, size_t bytes);

/* Fill a device (or host) buffer with Gen's samples, src/gen.rs:35-47 (device-side source
 * for `gen ... sparkfft` chains; A9). */
int qd_gen(const int64_t *cos_hz, size_t n_cos, uint64_t sample_rate, uint64_t first, size_t n,
           qd_c32 *out, int mem);

#ifdef __cplusplus
}
#endif
#endif
