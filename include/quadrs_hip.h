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

/* where a buffer lives.  QD_MEM_HOST_PINNED: host memory the HIP runtime can DMA from / to directly — allocated by
 * qd_host_alloc, registered by qd_host_register (e.g. an mmap of the file `from` opened, src/samples.rs:51-61), or
 * pinned by the caller's own HIP runtime; the host-resident path then skips its pageable -> pinned staging copy. */
typedef enum { QD_MEM_HOST = 0, QD_MEM_DEVICE = 1, QD_MEM_HOST_PINNED = 2 } qd_mem;

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

/* Stream (a hipStream_t, may be NULL) the fine-grained calls of THIS thread enqueue on; with QD_MEM_DEVICE buffers they
 * then return without synchronising (stream order is the only ordering), with host buffers they return after the
 * copy back.  Temporaries come from a process-wide workspace pool (no hipMalloc / hipFree per call);
 * qd_release_workspaces frees the pool's idle buffers and cached plans. */
int qd_set_stream(void *stream);
int qd_release_workspaces(void);

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
 * Any W: a power of two goes through the chain kernel's Radix4 (bit-exact against the oracle's restatement); every other
 * width up to 4096 (the front end's slider range, src/eui/mod.rs:157) through a Bluestein kernel carried in f64 — rustfft's
 * result for those lengths depends on its planner and host SIMD path (parity unpinned), so the bins are the exact DFT of
 * the windowed f32 samples rounded once to f32.
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
    int32_t  mode;            /* qd_mode: QD_MODE_EXACT (0, the default: the reference's products, order and roundings) or QD_MODE_FAST */
    int32_t  _pad2;
} qd_chain_desc;

/* QD_MODE_FAST is a PERMISSION, never the default and never what bench.py's `value` is measured in: the FIR may fuse each
 * multiply with its add (v_pk_fma_f32: one rounding per tap instead of two, same ascending-tap order), which breaks the
 * "within 1 ulp of the reference" bound of the exact mode — outputs stay within a few ulp of the window maximum of it and are, if
 * anything, closer to the infinitely precise filter.  Everything else (unpack, NCO, FFT, |X|) is unchanged.  Kernels without a
 * fused form (runtime-geometry kernels, overlapping windows, short filters) run the exact arithmetic: qd_plan_info.kernel_flags
 * bit 14 says whether the plan's kernel fuses. */
typedef enum { QD_MODE_EXACT = 0, QD_MODE_FAST = 1 } qd_mode;

typedef struct {
    uint64_t n_windows;       /* trip count of the sink's loop: spark_fft `while i < len - W`
                                 (src/fft.rs:28,65) or freq_levels `(len - W)/S` (src/fft.rs:86) */
    uint64_t decimated_len;   /* Samples::len() seen by the sink (LowPass::len, src/filter.rs:45-48) */
    uint64_t out_sample_rate; /* Samples::sample_rate() seen by the sink */
    uint64_t out_bytes_per_window;
    uint64_t raw_per_window;  /* W*D + T source samples one window reads */
    uint64_t raw_step;        /* S*D source samples between window starts */
    double   ratio;           /* Shift ratio (0 if no shift) */
    uint32_t tile_windows;    /* windows per workgroup tile (overlapping lowpass-free windows served by interleaved launches: any range; ranges starting on multiples of this keep the fast path) */
    uint32_t threads;         /* workgroup size */
    uint32_t lds_bytes;
    uint32_t kernel_kind;     /* 0 generic (runtime geometry), 1 built-in shape-specialised, 2 specialised at plan time (hiprtc) */
    uint32_t kernel_flags;    /* variant bits of the main kernel (0 for the generic kernels): 4 packed lane-per-output FIR, 8 row-aligned
                                 phase 1, 32 straight-line shared FIR, 64 deferred FFT, 128 packed two-output FIR, 256 non-temporal stream loads (65536: only rows no other tile reads), 8192 half-window tiles, 16384 fused FIR (QD_MODE_FAST),
                                 32768 three-stage kernel (producer / FIR / FFT waves on consecutive tiles), 131072 its streaming form (runs of tiles, state carried in LDS), 262144 the streaming kernel as the write sink (QD_EPI_CF32_BLOCKS: producers + FIR waves, no FFT stage),
                                 524288 the wave-local kernel of chains WITHOUT a lowpass (stride == width: one wave per tile, no workgroup barriers), 1048576 its form with the base butterflies out of the row registers (W = 128 ... 1024; plan-time builds), 2097152 overlapping windows of 2 ... 8 points, a window per lane (plan-time builds);
                                 chosen from the chain's geometry at plan time (built-in kernels: the same predicates, fixed at build time) */
    uint32_t _reserved;
} qd_plan_info;

/* How a plan picks its kernel and moves host-resident streams.  An explicit struct: the shipped library reads no tuning
 * environment variables (only the location of its on-disk code-object cache: QD_JIT_CACHE / XDG_CACHE_HOME / HOME). */
typedef enum {
    QD_KERNEL_AUTO = 0,          /* built-in shape-specialised kernel if one matches, else a cached / worthwhile plan-time build, else generic */
    QD_KERNEL_GENERIC = 1,       /* runtime-geometry kernels only */
    QD_KERNEL_SPECIALISE = 2,    /* always specialise at plan time (hiprtc) when no built-in kernel matches */
    QD_KERNEL_NO_PLAN_TIME = 3   /* built-in or generic; never hiprtc */
} qd_kernel_policy;

#define QD_MAX_SHARDS 16

typedef struct {
    uint32_t struct_size;        /* sizeof(qd_plan_options) */
    int32_t  kernel_policy;      /* qd_kernel_policy */
    int32_t  nco_order;          /* 0: chosen from |ratio|*n_samples; 1 / 2: force the first / second order NCO correction */
    uint32_t copy_threads;       /* host threads of the pageable -> pinned staging copy (QD_MEM_HOST); 0: up to 8 */
    uint64_t chunk_bytes;        /* source bytes per chunk of the host-resident path; 0: 64 MiB */
    uint32_t n_shards;           /* qd_plan_run_sharded*: number of window-range shards; 0 or 1: the current device only */
    int32_t  shard_device[QD_MAX_SHARDS];   /* HIP device of shard g (a device may serve several shards) */
    uint32_t tile_hint[8];       /* tuning: force a plan-time build with this tiling — windows per tile, threads (256 /
                                    512 / 1024), FIR outputs per lane, FIR block taps, waves per SIMD the build is register-
                                    budgeted for, LDS pad elements per row (1 / 2), FFT slots (tiles per FFT batch; 2 with the
                                    deferred FFT) in bits 0-7 of slot 6 and the kernel variant bits in bits 8-15 (1 planar LDS
                                    tile, 2 taps baked into the code, 4 packed lane-per-output FIR, 8 row-aligned phase 1, 16
                                    packed span FIR, 32 straight-line shared FIR, 64 deferred FFT, 128 packed two-output tile FIR;
                                    a bit a shape cannot take is ignored), workgroups per CU; all 0: the library's own choice.
                                    Every variant of one workgroup size computes the same bytes; with a shift stage, tilings of different workgroup
                                    size may round ~1e-8 of the NCO multipliers the other way (DESIGN.md sections 3.1, 4) */
} qd_plan_options;

int qd_plan_create(const qd_chain_desc *desc, qd_plan **plan);
int qd_plan_create_ex(const qd_chain_desc *desc, const qd_plan_options *options, qd_plan **plan);
int qd_plan_destroy(qd_plan *plan);
int qd_plan_get_info(const qd_plan *plan, qd_plan_info *info);
/* taps the plan designed (T floats), for inspection */
int qd_plan_get_taps(const qd_plan *plan, float *taps, size_t cap);
/* the main kernel's name as a profiler lists it (template name with its geometry, e.g. "qd::k_chain_pipe3s<0, 1, FixedGeo<64, 16, 32, 400, 14, ...>, ...>"),
 * NUL-terminated into buf[cap]; reporting only (bench.py's roofline.kernel) */
int qd_plan_kernel_name(const qd_plan *plan, char *buf, size_t cap);

/* source samples [*first, *first + *count) that windows [first_window, first_window+n) read */
int qd_plan_src_range(const qd_plan *plan, uint64_t first_window, uint64_t n_windows,
                      uint64_t *first, uint64_t *count);

/* Run windows [first_window, first_window + n_windows) of the sink's loop.
 * src holds the raw bytes of source samples [src_first, src_first + src_count) (a slab of the
 * stream; absolute sample indices keep the NCO phase identical to a whole-stream run).
 * out receives n_windows * out_bytes_per_window bytes.
 * Device buffers: the kernels are enqueued on `stream` (a hipStream_t, may be NULL) and the
 * call returns without synchronising.  Host buffers: chunked, double-buffered
 * hipMemcpyAsync in and out; returns after the last copy completed.  QD_MEM_HOST goes through a
 * pinned staging ring (a multi-threaded memcpy each way); QD_MEM_HOST_PINNED is copied from / to
 * directly.  src and out may be HOST and HOST_PINNED in any combination. */
int qd_plan_run(qd_plan *plan, const void *src, int src_mem, uint64_t src_first, uint64_t src_count,
                uint64_t first_window, uint64_t n_windows, void *out, int out_mem, void *stream);

/* Multi-GPU in one process (no Python, no collective).  The sink's windows are split into n_shards contiguous,
 * tile-aligned ranges (the reference's sink loop src/fft.rs:28-65 has no cross-window state); shard g runs on
 * options.shard_device[g] on its own streams, and the outputs concatenate to exactly the bytes of a one-device run.
 *   qd_plan_shard_info     window range and source range of shard g: it OWNS samples [own_first, +own_count) (disjoint,
 *                          in order) and additionally reads `halo` = (W-S)*D+T samples that the next shard owns.
 *   qd_plan_run_sharded    host-resident stream (QD_MEM_HOST / QD_MEM_HOST_PINNED): every shard's chunks, halo included,
 *                          are copied straight from the host buffer — the "host-side halo" of SURVEY section 5; one
 *                          host thread per shard drives that device's double-buffered ring.  Returns when all are done.
 *   qd_plan_run_sharded_device  device-resident, pre-split stream: slabs[g] is a buffer ON shard g's device that holds
 *                          the samples shard g owns and has room for `halo` more behind them; the halo is fetched from
 *                          the next shard's slab with hipMemcpyPeerAsync (xGMI when the devices differ), then the chain
 *                          runs; outs[g] (on the same device) receives that shard's windows.  sync != 0 waits for all.
 */
typedef struct {
    uint64_t w0, w1;             /* windows [w0, w1) of the sink's loop */
    uint64_t own_first, own_count, halo;
    int32_t  device;
    int32_t  _pad;
} qd_shard_info;
int qd_plan_shard_info(const qd_plan *plan, uint32_t shard, qd_shard_info *info);
int qd_plan_run_sharded(qd_plan *plan, const void *src, int src_mem, void *out, int out_mem);
int qd_plan_run_sharded_device(qd_plan *plan, void *const *slabs, void *const *outs, int sync);

/* Host-side figures of the most recent host-resident run of the plan (qd_plan_run with host buffers, or one shard of
 * qd_plan_run_sharded): the survey's qd_plan_stats. */
typedef struct {
    double   wall_ms;            /* whole call */
    double   stage_ms;           /* host time in the pageable <-> pinned staging copies (0 for QD_MEM_HOST_PINNED) */
    uint64_t bytes_h2d, bytes_d2h;
    uint32_t chunks;
    uint32_t _pad;
} qd_plan_stats;
int qd_plan_get_stats(const qd_plan *plan, qd_plan_stats *stats);

/* Pinned host memory for QD_MEM_HOST_PINNED: allocate, or register memory the caller already has (an mmap'ed file). */
int qd_host_alloc(size_t bytes, void **ptr);
int qd_host_free(void *ptr);
int qd_host_register(void *ptr, size_t bytes);
int qd_host_unregister(void *ptr);

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
