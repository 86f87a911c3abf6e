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
#include <mutex>
#include <thread>
#include <sys/stat.h>
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

uint32_t ilog2(uint64_t v) { uint32_t l = 0; while ((1ull << l) < v) ++l; return l; }
bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }

int spl_of(int fmt) { return fmt == QD_FMT_CF32 ? 2 : 4; }
int bps_of(int fmt) { return fmt == QD_FMT_CF32 ? 8 : (fmt == QD_FMT_CS16 ? 4 : 2); }

// ------------------------------------------------------------------ small kernels

// Row bases: (cos, sin)(fl((double)(row*ROW) * ratio)), plus theta and nf themselves.
__global__ void k_rowtab(double ratio, uint32_t row_len, uint64_t row0, uint64_t n_rows, RowBase *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    double nf = (double)((row0 + i) * (uint64_t)row_len);
    double th = nf * ratio;
    double s, c;
    sincos(th, &s, &c);
    RowBase rb;
    rb.c = c; rb.s = s; rb.theta = th; rb.nf = nf;
    out[i] = rb;
}

__global__ void k_jtab(double ratio, uint32_t n, double2 *out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double tj = (double)j * ratio;
    double s, c;
    sincos(tj, &s, &c);
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
        lr[u].jf = (double)j; lr[u].tj = lr[u].jf * ratio; lr[u].c = cs.x; lr[u].s = cs.y;
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
template <int F, int NCO, bool FI>
chain_fn pick_dyn(bool aligned) {
    return aligned ? k_chain<F, NCO, DynGeo, FI, 4, false, true, 4> : k_chain<F, NCO, DynGeo, FI, 4, false, false, 4>;
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

// ---- shape-specialised kernels (FixedGeo): the chain shapes of BASELINE.json / the README.
// Same source as the generic kernel with W,S,D,T,G as compile-time constants.
const FixedEntry kFixed[] = {
    // configs[1]  "shift 280000 | lowpass -power 20 -decimate 16 2000000 | sparkfft -width 128"   (README.md:57-63)
    QD_FIXED(0, 1, 128, 128, 16, 40, 2, 9, true, 4, "cfg2"),
    QD_FIXED(0, 2, 128, 128, 16, 40, 2, 9, true, 4, "cfg2"),
    // north_star target sentence: 200-tap FIR decimate 32 -> 128-pt FFT
    QD_FIXED(0, 1, 128, 128, 32, 200, 1, 9, true, 4, "cfg3p"),
    QD_FIXED(0, 2, 128, 128, 32, 200, 1, 9, true, 4, "cfg3p"),
    // README.md:90-94 / configs[2] (64-pt windows, stride 16, 400 taps): qd_longfir.hip
    // configs[3]  512-tap FIR decimate 8 -> 1024-pt FFT (no shift)
    // 70 KiB tile: one workgroup per CU, so give it 1024 threads (16 waves/CU); 5 rows of 2048 samples
    QD_FIXED_NT(0, 0, 1024, 1024, 8, 512, 1, 5, true, 4, 1024, 4, 2, 2, "cfg4"),
};

const FixedEntry *find_fixed(int fmt, int nco, uint32_t W, uint32_t S, uint32_t D, uint32_t T) {
    if (getenv("QD_NO_FIXED")) return nullptr;       // tests compare the specialised and generic kernels
    for (const FixedEntry &e : kFixed)
        if (e.fmt == fmt && e.nco == nco && e.W == W && e.S == S && e.D == D && e.T == T) return &e;
    int n_long = 0;
    const FixedEntry *lf = longfir_entries(&n_long);
    for (int i = 0; i < n_long; ++i)
        if (lf[i].fmt == fmt && lf[i].nco == nco && lf[i].W == W && lf[i].S == S && lf[i].D == D && lf[i].T == T) return &lf[i];
    return nullptr;
}

// ---- plan-time specialisation (hiprtc): any chain shape gets a FixedGeo build of the same kernel source.
// The headers are read from <dir of this .so>/csrc (the in-tree layout); compiled modules are cached per
// process.  QD_JIT=0 disables, QD_JIT=1 forces it for every plan; by default only streams whose chain
// input is >= 16 MiB pay the ~0.3 s compile.
struct JitKey {
    int fmt, nco, fir, rch, whole, lb, nt; uint32_t W, S, D, T, G; uint32_t firb = 8, firr = 1; int noslp = 0; uint32_t pad = 1;
    bool operator<(const JitKey &o) const {
        return std::tie(fmt, nco, fir, rch, whole, lb, nt, W, S, D, T, G, firb, firr, noslp, pad) <
               std::tie(o.fmt, o.nco, o.fir, o.rch, o.whole, o.lb, o.nt, o.W, o.S, o.D, o.T, o.G, o.firb, o.firr, o.noslp, o.pad);
    }
};
std::mutex g_jit_mu;
std::map<JitKey, hipFunction_t> g_jit_cache;

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
hipFunction_t jit_chain_kernel(const JitKey &k, std::string *why, bool may_compile = true) {
    std::lock_guard<std::mutex> lock(g_jit_mu);
    auto it = g_jit_cache.find(k);
    if (it != g_jit_cache.end()) return it->second;
    const std::string dir = csrc_dir();
    std::vector<char> hdr1, hdr2;
    if (!read_file(dir + "/qd_chain.h", &hdr1) || !read_file(dir + "/qd_device.h", &hdr2)) {
        *why = "kernel headers not found next to the library (" + dir + ")";
        return nullptr;
    }
    char name[512];
    snprintf(name, sizeof name, "qd::k_chain<%d, %d, qd::FixedGeo<%u, %u, %u, %u, %u, %u, %u, %u>, %s, %d, %s, true, %d, %d>", k.fmt, k.nco, k.W,
             k.S, k.D, k.T, k.G, k.firb, k.firr, k.pad, k.fir ? "true" : "false", k.rch, k.whole ? "true" : "false", k.lb, k.nt);
    std::string src = std::string("#include \"qd_chain.h\"\ntemplate __global__ void ") + name + "(const qd::ChainParams);\n";
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "qd_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *why = "hiprtcCreateProgram failed"; return nullptr; }
    hiprtcAddNameExpression(prog, name);
    const std::string inc = "-I" + dir;
#ifdef QD_STAMP
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", inc.c_str(), "-DQD_STAMP"};
#else
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", inc.c_str()};
#endif
    std::vector<const char *> optv(opts, opts + sizeof opts / sizeof opts[0]);
    if (k.noslp || getenv("QD_JIT_NOSLP")) optv.push_back("-fno-slp-vectorize");   // scalar f32 accumulate chains (see qd_longfir.hip)
    std::vector<std::string> extra;                          // development: QD_JIT_FLAGS="-mllvm -foo ..." appended verbatim
    if (const char *e = getenv("QD_JIT_FLAGS")) {
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
        for (const char *o : optv) if (o != inc.c_str()) h = fnv1a(o, strlen(o) + 1, h);
        h = fnv1a(hdr1.data(), hdr1.size(), h);
        h = fnv1a(hdr2.data(), hdr2.size(), h);
        const std::string cdir = getenv("QD_JIT_DUMP") ? std::string() : jit_cache_dir();
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
                    g_jit_cache[k] = fn;
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
        hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    const char *lowered = nullptr;
    hiprtcGetLoweredName(prog, name, &lowered);
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    if (const char *dump = getenv("QD_JIT_DUMP")) {          // development: keep the code object for llvm-objdump
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
    g_jit_cache[k] = fn;
    return fn;
}

struct Geometry {
    uint32_t G = 1, Dp = 1, lds_raw_elems = 0;
    size_t lds_bytes = 0;
};

size_t lds_for(uint32_t G, uint64_t W, uint64_t S, uint64_t D, uint64_t T, uint32_t *raw_elems, uint32_t pad_per_row = 1) {
    uint64_t tile_raw = (uint64_t)(G - 1) * S * D + W * D + T;
    uint64_t pad = (D % 2 == 0) ? pad_per_row * (tile_raw / D + 1) : 0;
    uint64_t elems = tile_raw + pad + 1;
    uint64_t min_elems = (uint64_t)G * W / 2 + 1;     // bucket epilogue parks G*W f32 norms here
    if (elems < min_elems) elems = min_elems;
    elems = (elems + 1) & ~1ull;                      // keep fb 16-byte aligned
    if (raw_elems) *raw_elems = (uint32_t)elems;
    uint64_t shared_fir = (T && S < W) ? 2 * ((uint64_t)(G - 1) * S + W) * 8 : 0;   // dec[] + trc[] of the shared-FIR mode
    return (size_t)(elems * 8 + (uint64_t)G * W * 8 + W * 8 + ((T + 3) & ~3ull) * 4 + 256 * 4 + shared_fir);   // raw tile + FFT buffer + twiddles + taps + 8-bit LUT (+ shared FIR)
}

constexpr size_t kLdsMax = 160 * 1024;

}  // namespace

// ------------------------------------------------------------------ plan

struct qd_plan {
    qd_chain_desc d{};
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
    hipFunction_t jit_fn = nullptr;      // plan-time specialised kernel (hiprtc), replaces fn for aligned launches
    std::string jit_note;
    int wg_per_cu = 1, n_cu = 256, prefetch_mode = 2, nco = 0, nt = kThreads;
    // NCO tables
    double2 *jtab_d = nullptr;
    RowBase *rowtab_d = nullptr;
    uint64_t rowtab_row0 = 0, rowtab_rows = 0;
    // plans whose main kernel uses another workgroup size keep a second pair of NCO tables laid out for the
    // 256-thread per-sample kernel that takes the windows at an unaligned slab end
    double2 *jtab256_d = nullptr;
    RowBase *rowtab256_d = nullptr;
    uint64_t rowtab256_row0 = 0, rowtab256_rows = 0;
    // take_fft mode (generic kernels): per-window start offsets and an f32 window, both on the device
    const uint64_t *row_offsets_d = nullptr;
    const float *window_d = nullptr;
    // timing
    bool timing = false, ev_made = false, ev_recorded = false;
    hipEvent_t ev0{}, ev1{};
    // host streaming
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
    void *dev_in[2] = {nullptr, nullptr}, *dev_out[2] = {nullptr, nullptr};
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    hipStream_t streams[2] = {nullptr, nullptr};
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

int ensure_rowtab_for(qd_plan *p, uint32_t ROW, RowBase **tab, uint64_t *row0, uint64_t *nrows, uint64_t n_lo, uint64_t n_hi,
                      hipStream_t st) {
    uint64_t r_lo = n_lo / ROW, r_hi = (n_hi + ROW - 1) / ROW + 1;
    if (*tab && r_lo >= *row0 && r_hi <= *row0 + *nrows) return QD_OK;
    if (*tab) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(*tab)); *tab = nullptr; }
    uint64_t rows = r_hi - r_lo;
    HIPCHK(hipMalloc(tab, rows * sizeof(RowBase)));
    *row0 = r_lo; *nrows = rows;
    uint32_t blocks = (uint32_t)((rows + 255) / 256);
    hipLaunchKernelGGL(k_rowtab, dim3(blocks), dim3(256), 0, st, p->ratio, ROW, r_lo, rows, *tab);
    HIPCHK(hipGetLastError());
    return QD_OK;
}

int ensure_rowtab(qd_plan *p, uint64_t n_lo, uint64_t n_hi, hipStream_t st) {
    if (!p->has_shift) return QD_OK;
    return ensure_rowtab_for(p, p->nt * spl_of(p->d.format), &p->rowtab_d, &p->rowtab_row0, &p->rowtab_rows, n_lo, n_hi, st);
}

int launch_chain(qd_plan *p, const void *src_d, uint64_t src_first, uint64_t src_count, uint64_t first_window,
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
    int rc = ensure_rowtab(p, need0, need1, st);
    if (rc) return rc;

    ChainParams P{};
    P.src = static_cast<const uint8_t *>(src_d);
    P.src_first = src_first; P.src_count = src_count;
    P.out_window0 = out_window0;
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
    P.root2 = (float)std::sqrt(0.5);
    P.tw16_1 = compute_twiddle(1, 16); P.tw16_2 = compute_twiddle(2, 16); P.tw16_3 = compute_twiddle(3, 16);
    P.ratio = p->ratio;
    P.rowtab = p->rowtab_d; P.rowtab_row0 = p->rowtab_row0;
    P.jtab = p->jtab_d; P.taps = p->taps_d; P.tw = p->tw_d;
    P.out = out_d;
    P.row_offsets = p->row_offsets_d; P.window = p->window_d;
    P.blk_len = p->blk_len ? p->blk_len : p->W; P.blk_sub_mask = p->blk_subs - 1;
    P.tile_extra = p->tile_extra;
    if (const char *e = getenv("QD_DEBUG_SKIP")) P.dbg = (uint32_t)atoi(e);   // timing-only ablation, never set in tests/bench
#ifdef QD_STAMP
    static unsigned long long *stamps_d = nullptr;
    if (!stamps_d) { HIPCHK(hipMalloc(&stamps_d, 160 * 8)); }
    HIPCHK(hipMemsetAsync(stamps_d, 0, 160 * 8, st));
    P.stamps = stamps_d;
#endif

    // The aligned kernels issue whole-vector loads: the slab must start on a vector boundary and a
    // window whose last vector would straddle the slab end goes to the per-sample kernel instead.
    const int vec_bytes = spl * bps;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(src_d) % vec_bytes) == 0 && (src_first % spl) == 0 &&
                        src_count * (uint64_t)bps >= (uint64_t)vec_bytes;
    uint64_t n_aligned = 0;
    if (vec_ok) {
        // windows [first_window, first_window + n_aligned): need-end rounded up to a vector fits in the slab
        const uint64_t step = (uint64_t)p->S * p->D, rpw = (uint64_t)p->W * p->D + p->T;
        const uint64_t usable = (src_count / spl) * spl + src_first;     // end of the last whole vector
        n_aligned = n_windows;
        while (n_aligned > 0 && (first_window + n_aligned - 1) * step + rpw > usable) --n_aligned;
    }
    const bool tail_tables = n_aligned < n_windows && p->nt != kThreads && p->has_shift;
    if (tail_tables) {
        const uint64_t t0 = (first_window + n_aligned) * p->S * p->D;
        rc = ensure_rowtab_for(p, kThreads * spl, &p->rowtab256_d, &p->rowtab256_row0, &p->rowtab256_rows, t0, need1, st);
        if (rc) return rc;
    }
    uint64_t cap = (uint64_t)p->n_cu * p->wg_per_cu;
    if (const char *e = getenv("QD_WG_PER_CU")) cap = (uint64_t)p->n_cu * (uint64_t)atoi(e);   // tuning knob
    if (p->timing) {
        if (!p->ev_made) { HIPCHK(hipEventCreate(&p->ev0)); HIPCHK(hipEventCreate(&p->ev1)); p->ev_made = true; }
        HIPCHK(hipEventRecord(p->ev0, st));
    }
    for (int part = 0; part < 2; ++part) {
        const uint64_t w_begin = part == 0 ? first_window : first_window + n_aligned;
        const uint64_t w_count = part == 0 ? n_aligned : n_windows - n_aligned;
        if (w_count == 0) continue;
        P.first_window = w_begin; P.n_windows = w_count;
        if (part == 1 && tail_tables) { P.rowtab = p->rowtab256_d; P.rowtab_row0 = p->rowtab256_row0; P.jtab = p->jtab256_d; }
        const uint64_t n_tiles = (w_count + P.G - 1) / P.G;
        const uint32_t grid = (uint32_t)(n_tiles < cap ? n_tiles : cap);
        if (part == 0 && p->jit_fn && !p->row_offsets_d) {
            void *args[] = {&P};
            HIPCHK(hipModuleLaunchKernel(p->jit_fn, grid, 1, 1, (unsigned)p->nt, 1, 1, (unsigned)p->geo.lds_bytes, st, args, nullptr));
        } else {
            hipLaunchKernelGGL(part == 0 ? p->fn : p->fn_unaligned, dim3(grid), dim3(part == 0 ? p->nt : kThreads), p->geo.lds_bytes, st, P);
            HIPCHK(hipGetLastError());
        }
    }
    if (p->timing) { HIPCHK(hipEventRecord(p->ev1, st)); p->ev_recorded = true; }
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
    p->stage_in_bytes = p->stage_out_bytes = 0;
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

static int plan_init(qd_plan *p, const qd_chain_desc &d, uint64_t len, uint64_t rate) {
    (void)hipGetDevice(&p->device);
    p->has_shift = d.has_shift != 0;
    p->has_fir = d.has_lowpass != 0;
    p->W = (uint32_t)d.width; p->logW = ilog2(d.width); p->S = (uint32_t)d.stride;
    if (d.epilogue == QD_EPI_CF32_BLOCKS) {          // tiles are sub-blocks of <= 256 outputs of a read_at block of d.width
        p->blk_len = (uint32_t)d.width;
        p->W = p->blk_len < 256 ? p->blk_len : 256;
        p->logW = ilog2(p->W); p->S = p->W;
        p->blk_subs = p->blk_len / p->W;
        const uint32_t c = (uint32_t)(d.taps - d.taps / 2);
        p->tile_extra = c > d.decimate ? c - (uint32_t)d.decimate : 0;
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
    // second-order term is e^2/2 with |e| <= 1.5 ulp(place); below 2^27 rad that is <= 2.5e-16, inside the
    // scheme's ~4e-16 error budget, above it the second-order kernel is used
    p->nco = !p->has_shift ? 0 : ((std::fabs(p->ratio) * (double)d.n_samples > 134217728.0) ? 2 : 1);
    if (const char *e = getenv("QD_NCO_ORDER")) { int v = atoi(e); if (p->has_shift && (v == 1 || v == 2)) p->nco = v; }

    // tile geometry: a shape-specialised kernel dictates G; otherwise pick G for LDS / lane use
    uint32_t G = 1, raw_elems = 0;
    const uint32_t T_lds = p->T + p->tile_extra;     // LDS sizing sees the extended tile
    if (lds_for(1, p->W, p->S, p->D, T_lds, &raw_elems) > kLdsMax)
        return fail(QD_ERR_UNSUPPORTED, "one window (W*D+T = %llu samples) exceeds the 160 KiB LDS tile",
                    (unsigned long long)((uint64_t)d.width * (d.has_lowpass ? d.decimate : 1) + (d.has_lowpass ? d.taps : 0)));
    p->fixed = (p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS) ? find_fixed(d.format, p->nco, p->W, p->S, p->D, p->T) : nullptr;
    // QD_TUNE=G:NT:FIRR:FIRB:LB:PAD (development): force a plan-time build with this tiling instead of the table / heuristics
    // (LB = waves per SIMD the build is register-budgeted for: 4 -> 128 VGPRs, 2 -> 256; PAD = LDS pad elements per row)
    uint32_t tune[6] = {0, 0, 1, 8, 4, 1};
    bool tuned = false;
    if (const char *e = getenv("QD_TUNE")) {
        if (p->has_fir && d.epilogue != QD_EPI_CF32_BLOCKS && sscanf(e, "%u:%u:%u:%u:%u:%u", &tune[0], &tune[1], &tune[2], &tune[3], &tune[4], &tune[5]) >= 2 && tune[4] >= 1 && tune[4] <= 8 &&
            tune[0] >= 1 && (tune[1] == 256 || tune[1] == 512 || tune[1] == 1024) && (tune[5] == 1 || tune[5] == 2) &&
            lds_for(tune[0], p->W, p->S, p->D, T_lds, nullptr, tune[5]) <= kLdsMax) {
            tuned = true;
            p->fixed = nullptr;
        }
    }
    // Plan-time specialisation is wanted for shapes without a built-in kernel once the stream is big enough to
    // repay the ~0.3 s compile (QD_JIT=0 off, =1 always).
    const char *jenv = getenv("QD_JIT");
    const int jmode = jenv ? atoi(jenv) : -1;                       // -1 auto, 0 off, 1 force
    const uint64_t in_bytes = (uint64_t)d.n_samples * bps_of(d.format);
    const bool jit_ok = !getenv("QD_NO_FIXED") && d.epilogue != QD_EPI_CF32_BLOCKS && jmode != 0;
    // a cached build is always used; a NEW build only when forced or when the stream is at least 1 GiB
    const bool may_compile = jmode == 1 || tuned || in_bytes >= (1ull << 30);
    auto make_key = [&](uint32_t g, int nt, int lb, int noslp, uint32_t padv) {
        const uint64_t ROW = (uint64_t)nt * spl_of(d.format);
        const uint64_t tile_raw = (uint64_t)(g - 1) * p->S * p->D + (uint64_t)p->W * p->D + p->T;
        // a run may start at any window, so a tile starts on a row boundary only if S*D is a multiple of ROW
        const uint64_t rows = (tile_raw + ROW - 1) / ROW + ((((uint64_t)p->S * p->D) % ROW) ? 1 : 0);
        return JitKey{d.format, p->nco, p->has_fir ? 1 : 0, rows <= 10 ? (int)rows : 4, rows <= 10 ? 1 : 0, lb, nt,
                      p->W, p->S, p->D, p->T, g, tune[3], tune[2], noslp, padv};
    };
    // FIR-dominated shapes (>= 8 taps per input sample): a tile's FIR phase is latency-bound — one wave walks
    // all T taps however few outputs the tile has — so take the largest tile with <= 512 FIR outputs that LDS
    // allows, 512 threads, a 256-VGPR budget and scalar accumulate chains (measured 1.3-2.9x over the small-tile
    // default on six such shapes, scripts/policy_probe.py)
    bool heavy = jit_ok && !tuned && !p->fixed && p->has_fir && (uint64_t)p->T >= 8ull * p->D;
    int jit_lb = 4, jit_noslp = 0;
    uint32_t pad = 1;              // LDS pad elements per row the main kernel is built with (FixedGeo PAD_)
    if (heavy) {
        auto outs = [&](uint32_t g) { return p->S < p->W ? (uint64_t)(g - 1) * p->S + p->W : (uint64_t)g * p->W; };
        uint32_t gh = 1;           // 16-byte aligned LDS rows (pad 2): ds_read_b128 sample pairs in the tap loop (FixedGeo::kPad)
        while (gh < 64 && outs(gh + 1) <= 512 && lds_for(gh + 1, p->W, p->S, p->D, T_lds, nullptr, 2) <= kLdsMax) ++gh;
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
    } else if (p->fixed) {
        G = p->fixed->G;
        p->nt = p->fixed->nt;
        pad = (uint32_t)p->fixed->pad;
    } else {
        while (G < 64 && (uint64_t)G * p->W < 256 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr) <= 40 * 1024) G *= 2;
        while (G < 64 && (uint64_t)G * p->W < 1024 && lds_for(G * 2, p->W, p->S, p->D, T_lds, nullptr) <= 36 * 1024) G *= 2;
        if (p->n_windows && G > p->n_windows) { while (G > 1 && G / 2 >= p->n_windows) G /= 2; }
    }
    p->geo.G = G;
    p->geo.lds_bytes = lds_for(G, p->W, p->S, p->D, T_lds, &raw_elems, pad);     // the generic kernels (pad 1) fit inside the same allocation
    p->geo.lds_raw_elems = raw_elems;
    p->geo.Dp = p->D + ((p->D % 2 == 0) ? 1 : 0);
    if ((uint64_t)raw_elems * p->D >= (1ull << 32)) return fail(QD_ERR_UNSUPPORTED, "tile too large");

    p->fn = p->fixed ? p->fixed->fn : pick_generic(d.format, p->nco, p->has_fir, true);
    p->fn_unaligned = pick_generic(d.format, p->nco, p->has_fir, false);
    if (!p->fn || !p->fn_unaligned) return fail(QD_ERR_UNSUPPORTED, "no kernel built for this format (QD_DEV_FAST build?)");
    {
        // plan-time specialisation for shapes without a built-in FixedGeo kernel
        const bool want = !heavy && (tuned || (!p->fixed && jit_ok));
        if (want) {
            p->jit_fn = jit_chain_kernel(make_key(G, p->nt, jit_lb, jit_noslp, pad), &p->jit_note, may_compile);
            if (tuned && !p->jit_fn) return fail(QD_ERR_UNSUPPORTED, "QD_TUNE build failed: %s", p->jit_note.c_str());
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) p->n_cu = prop.multiProcessorCount;
    int by_lds = (int)(kLdsMax / p->geo.lds_bytes);
    p->wg_per_cu = by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
    if (p->fixed) { int by_regs = p->fixed->lb * 256 / p->fixed->nt; if (by_regs < 1) by_regs = 1; if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }
    if (p->nt > kThreads) { int by_threads = 2048 / p->nt; if (p->wg_per_cu > by_threads) p->wg_per_cu = by_threads; }
    if (tuned || heavy) { int by_regs = (jit_lb * 4 * 64) / p->nt; if (by_regs < 1) by_regs = 1; if (p->wg_per_cu > by_regs) p->wg_per_cu = by_regs; }
    if (p->jit_fn && p->geo.lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(p->jit_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->geo.lds_bytes) != hipSuccess)
            p->jit_fn = nullptr;     // fall back to the generic kernel
    }
    if (heavy && !p->jit_fn) p->nt = kThreads;
    for (chain_fn f : {p->fn, p->fn_unaligned}) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p->geo.lds_bytes);
        if (e != hipSuccess)
            return fail(QD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS %zu): %s", p->geo.lds_bytes, hipGetErrorString(e));
    }

    // constant tables
    p->fft = fft_layout(d.width);
    if (!p->fft.tw.empty()) {
        HIPCHK(hipMalloc(&p->tw_d, p->fft.tw.size() * sizeof(float2)));
        HIPCHK(hipMemcpy(p->tw_d, p->fft.tw.data(), p->fft.tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (p->has_fir) {
        p->taps_h.resize(p->T);
        design_taps(d.lowpass_hz, d.sample_rate, p->T, p->taps_h.data());
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


int qd_plan_create(const qd_chain_desc *desc, qd_plan **out) {
    if (!desc || !out) return fail(QD_ERR_INVALID, "desc/plan is NULL");
    if (desc->struct_size != sizeof(qd_chain_desc)) return fail(QD_ERR_INVALID, "qd_chain_desc size mismatch");
    const qd_chain_desc &d = *desc;
    if (d.format < 0 || d.format > 3) return fail(QD_ERR_INVALID, "unknown format %d", d.format);
    if (d.epilogue < 0 || d.epilogue > 3) return fail(QD_ERR_INVALID, "unknown epilogue %d", d.epilogue);
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
    qd_plan *p = new qd_plan();
    p->d = d;
    int rc = plan_init(p, d, len, rate);
    if (rc) { qd_plan_destroy(p); return rc; }
    *out = p;
    return QD_OK;
}

int qd_plan_destroy(qd_plan *p) {
    if (!p) return QD_OK;
    (void)hipDeviceSynchronize();
    free_streaming(p);
    if (p->taps_d) (void)hipFree(p->taps_d);
    if (p->tw_d) (void)hipFree(p->tw_d);
    if (p->jtab_d) (void)hipFree(p->jtab_d);
    if (p->rowtab_d) (void)hipFree(p->rowtab_d);
    if (p->jtab256_d) (void)hipFree(p->jtab256_d);
    if (p->rowtab256_d) (void)hipFree(p->rowtab256_d);
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
    info->tile_windows = p->geo.G;
    info->threads = (uint32_t)p->nt;
    info->lds_bytes = (uint32_t)p->geo.lds_bytes;
    info->kernel_kind = p->jit_fn ? 2u : (p->fixed ? 1u : 0u);
    return QD_OK;
}

int qd_plan_get_taps(const qd_plan *p, float *taps, size_t cap) {
    if (!p || !taps) return fail(QD_ERR_INVALID, "plan/taps is NULL");
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
    return QD_OK;
}

int qd_plan_last_kernel_ms(qd_plan *p, float *ms) {
    if (!p || !ms) return fail(QD_ERR_INVALID, "NULL argument");
    if (!p->ev_recorded) return fail(QD_ERR_INVALID, "no timed run recorded");
    HIPCHK(hipEventSynchronize(p->ev1));
    HIPCHK(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return QD_OK;
}

namespace {
// Pageable -> pinned staging copy on several host threads: one thread moves ~10-15 GB/s, which would cap the
// host-resident path far below PCIe; QD_COPY_THREADS overrides the thread count (default: up to 8).
void par_memcpy(void *dst, const void *src, size_t n) {
    static const unsigned n_thr = [] {
        if (const char *e = getenv("QD_COPY_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) return (unsigned)v; }
        unsigned hw = std::thread::hardware_concurrency();
        unsigned t = hw / 2; if (t < 1) t = 1; if (t > 8) t = 8;
        return t;
    }();
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (src_mem == QD_MEM_DEVICE && out_mem == QD_MEM_DEVICE)
        return launch_chain(p, src, src_first, src_count, first_window * subs, n_windows * subs, first_window * subs, out, st);
    if (src_mem != QD_MEM_HOST || out_mem != QD_MEM_HOST)
        return fail(QD_ERR_UNSUPPORTED, "mixed host/device buffers are not supported; use both host or both device");

    // host-resident stream: chunked, double-buffered H2D / kernel / D2H
    const int bps = bps_of(p->d.format);
    const uint64_t obw = out_bytes_per_window(p) / subs;      // per kernel window (a sub-block for QD_EPI_CF32_BLOCKS)
    first_window *= subs; n_windows *= subs;                  // from here on: kernel-window units
    const uint64_t step = (uint64_t)p->S * p->D, rpw = (uint64_t)p->W * p->D + p->T;
    uint64_t target_bytes = 64ull << 20;
    if (const char *e = getenv("QD_CHUNK_MB")) { int v = atoi(e); if (v >= 1 && v <= 4096) target_bytes = (uint64_t)v << 20; }   // tuning knob
    uint64_t cw = target_bytes / (step * bps ? step * bps : 1);
    if (cw < p->geo.G) cw = p->geo.G;
    cw = (cw / p->geo.G) * p->geo.G;
    if (cw > n_windows) cw = n_windows ? n_windows : 1;
    const size_t in_bytes = (size_t)(((cw - 1) * step + rpw + 8) * bps), ob = (size_t)(cw * obw);
    if (in_bytes > p->stage_in_bytes || ob > p->stage_out_bytes) {
        free_streaming(p);
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc(&p->pin_in[i], in_bytes, hipHostMallocDefault));
            HIPCHK(hipHostMalloc(&p->pin_out[i], ob, hipHostMallocDefault));
            HIPCHK(hipMalloc(&p->dev_in[i], in_bytes));
            HIPCHK(hipMalloc(&p->dev_out[i], ob));
            HIPCHK(hipStreamCreateWithFlags(&p->streams[i], hipStreamNonBlocking));
        }
        p->stage_in_bytes = in_bytes; p->stage_out_bytes = ob;
    }
    struct Pending { bool live = false; uint64_t w0 = 0, nw = 0; } pend[2];
    auto drain = [&](int slot) -> int {
        if (!pend[slot].live) return QD_OK;
        HIPCHK(hipStreamSynchronize(p->streams[slot]));
        par_memcpy(static_cast<uint8_t *>(out) + (pend[slot].w0 - first_window) * obw, p->pin_out[slot], pend[slot].nw * obw);
        pend[slot].live = false;
        return QD_OK;
    };
    int slot = 0;
    for (uint64_t w = first_window; w < first_window + n_windows; w += cw, slot ^= 1) {
        int rc = drain(slot);
        if (rc) return rc;
        uint64_t nw = first_window + n_windows - w < cw ? first_window + n_windows - w : cw;
        uint64_t s0 = w * step, cnt = (nw - 1) * step + rpw;
        // keep vector loads aligned: start the slab on a multiple of 8 samples
        uint64_t s0a = s0 & ~7ull;
        if (s0a < src_first) s0a = src_first;
        uint64_t cnta = s0 + cnt - s0a;
        if (s0a < src_first || s0a + cnta > src_first + src_count)
            return fail(QD_ERR_INVALID, "src slab does not cover the requested windows");
        par_memcpy(p->pin_in[slot], static_cast<const uint8_t *>(src) + (s0a - src_first) * bps, cnta * bps);
        HIPCHK(hipMemcpyAsync(p->dev_in[slot], p->pin_in[slot], cnta * bps, hipMemcpyHostToDevice, p->streams[slot]));
        rc = launch_chain(p, p->dev_in[slot], s0a, cnta, w, nw, w, p->dev_out[slot], p->streams[slot]);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(p->pin_out[slot], p->dev_out[slot], nw * obw, hipMemcpyDeviceToHost, p->streams[slot]));
        pend[slot].live = true; pend[slot].w0 = w; pend[slot].nw = nw;
    }
    int rc = drain(0);
    if (rc) return rc;
    return drain(1);
}

// ------------------------------------------------------------------ fine-grained ops

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

int qd_unpack(int fmt, const void *bytes, size_t n_pairs, qd_c32 *out, int mem) {
    if (fmt < 0 || fmt > 3) return fail(QD_ERR_INVALID, "unknown format %d", fmt);
    if (n_pairs == 0) return QD_OK;
    if (!bytes || !out) return fail(QD_ERR_INVALID, "NULL buffer");
    const size_t ib = n_pairs * qd_pair_bytes(fmt), ob = n_pairs * 8;
    DevBuf di, dout;
    const void *src = bytes; void *dst = out;
    if (mem == QD_MEM_HOST) {
        HIPCHK(hipMalloc(&di.p, ib)); HIPCHK(hipMalloc(&dout.p, ob));
        HIPCHK(hipMemcpy(di.p, bytes, ib, hipMemcpyHostToDevice));
        src = di.p; dst = dout.p;
    }
    size_t blocks = (n_pairs + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_unpack, dim3((uint32_t)blocks), dim3(256), 0, 0, fmt, static_cast<const uint8_t *>(src), n_pairs,
                       static_cast<float2 *>(dst));
    HIPCHK(hipGetLastError());
    if (mem == QD_MEM_HOST) HIPCHK(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
    return QD_OK;
}

int qd_shift(qd_c32 *buf, size_t n, uint64_t abs_off, double ratio, int mem) {
    if (n == 0) return QD_OK;
    if (!buf) return fail(QD_ERR_INVALID, "NULL buffer");
    constexpr uint32_t ROW = 512;
    DevBuf db, rt, jt;
    float2 *d = reinterpret_cast<float2 *>(buf);
    if (mem == QD_MEM_HOST) {
        HIPCHK(hipMalloc(&db.p, n * 8));
        HIPCHK(hipMemcpy(db.p, buf, n * 8, hipMemcpyHostToDevice));
        d = static_cast<float2 *>(db.p);
    }
    uint64_t r0 = abs_off / ROW, r1 = (abs_off + n + ROW - 1) / ROW, rows = r1 - r0;
    HIPCHK(hipMalloc(&rt.p, rows * sizeof(RowBase)));
    HIPCHK(hipMalloc(&jt.p, ROW * sizeof(double2)));
    hipLaunchKernelGGL(k_rowtab, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, 0, ratio, ROW, r0, rows, static_cast<RowBase *>(rt.p));
    hipLaunchKernelGGL(k_jtab, dim3(2), dim3(256), 0, 0, ratio, ROW, static_cast<double2 *>(jt.p));
    int so = (std::fabs(ratio) * (double)(abs_off + n) > 134217728.0) ? 1 : 0;
    uint32_t grid = (uint32_t)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(k_shift, dim3(grid), dim3(256), 0, 0, d, abs_off, (uint64_t)n, ratio, static_cast<const RowBase *>(rt.p), r0, rows,
                       static_cast<const double2 *>(jt.p), so);
    HIPCHK(hipGetLastError());
    if (mem == QD_MEM_HOST) HIPCHK(hipMemcpy(buf, db.p, n * 8, hipMemcpyDeviceToHost));
    else HIPCHK(hipDeviceSynchronize());   // tables are freed on return
    return QD_OK;
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
    DevBuf dt, dr, dout;
    HIPCHK(hipMalloc(&dt.p, T * 4));
    HIPCHK(hipMemcpy(dt.p, taps, T * 4, hipMemcpyHostToDevice));   // taps are always host (O(T))
    const float2 *r = reinterpret_cast<const float2 *>(raw);
    float2 *o = reinterpret_cast<float2 *>(out);
    if (mem == QD_MEM_HOST) {
        HIPCHK(hipMalloc(&dr.p, valid * 8)); HIPCHK(hipMalloc(&dout.p, out_n * 8));
        HIPCHK(hipMemcpy(dr.p, raw, valid * 8, hipMemcpyHostToDevice));
        r = static_cast<const float2 *>(dr.p); o = static_cast<float2 *>(dout.p);
    }
    size_t blocks = (out_n + 127) / 128; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_lowpass_block, dim3((uint32_t)blocks), dim3(128), 0, 0, static_cast<const float *>(dt.p), (uint32_t)T, D, r,
                       (uint64_t)valid, o, (uint64_t)out_n);
    HIPCHK(hipGetLastError());
    if (mem == QD_MEM_HOST) HIPCHK(hipMemcpy(out, dout.p, out_n * 8, hipMemcpyDeviceToHost));
    else HIPCHK(hipDeviceSynchronize());
    return QD_OK;
}

int qd_fft_norm_batch(const qd_c32 *in, size_t W, size_t n_fft, size_t in_stride, float *norms, int mem) {
    if (n_fft == 0) return QD_OK;
    if (!in || !norms) return fail(QD_ERR_INVALID, "NULL buffer");
    if (in_stride == 0) return fail(QD_ERR_INVALID, "in_stride 0");
    qd_chain_desc d{};
    d.struct_size = sizeof d;
    d.format = QD_FMT_CF32; d.sample_rate = 1;
    d.n_samples = (n_fft - 1) * in_stride + W + 1;     // so that the strict `<` loop yields n_fft windows
    d.width = W; d.stride = in_stride; d.epilogue = QD_EPI_NORMS_F32;
    // the window loop `i < len - W` needs len - W > (n_fft-1)*stride: len = (n_fft-1)*stride + W + 1
    qd_plan *p = nullptr;
    int rc = qd_plan_create(&d, &p);
    if (rc) return rc;
    uint64_t have = (n_fft - 1) * in_stride + W;
    rc = qd_plan_run(p, in, mem, 0, have, 0, n_fft, norms, mem, nullptr);
    if (rc == QD_OK && mem == QD_MEM_DEVICE) { if (hipDeviceSynchronize() != hipSuccess) rc = fail(QD_ERR_HIP, "sync failed"); }
    qd_plan_destroy(p);
    return rc;
}

int qd_take_fft(const qd_c32 *in, uint64_t in_first, size_t n_in, uint64_t samples_len, int has_slice,
                uint64_t start, uint64_t end, size_t W, int windowing, size_t output_len, float *rows, int mem) {
    if (!in || !rows) return fail(QD_ERR_INVALID, "NULL buffer");
    if (!is_pow2(W)) return fail(QD_ERR_UNSUPPORTED, "take_fft width %zu: only power-of-two widths are built (the reference's planner takes any)", W);
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
    qd_chain_desc d{};
    d.struct_size = sizeof d;
    d.format = QD_FMT_CF32; d.sample_rate = 1;
    d.n_samples = (uint64_t)output_len * W + 1;        // only sizes the sink loop: output_len windows at stride W
    d.width = W; d.stride = W; d.epilogue = QD_EPI_NORMS_F32;
    qd_plan *p = nullptr;
    int rc = qd_plan_create(&d, &p);
    if (rc) return rc;
    DevBuf doffs, dwin, din, dout;
    auto cleanup = [&](int r) { qd_plan_destroy(p); return r; };
    p->geo.G = 1;                                       // one irregular row per tile
    {
        uint32_t raw_elems = 0;
        p->geo.lds_bytes = lds_for(1, p->W, p->S, p->D, p->T, &raw_elems);
        p->geo.lds_raw_elems = raw_elems;
    }
    if (hipMalloc(&doffs.p, output_len * 8) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "hipMalloc offsets"));
    if (hipMemcpy(doffs.p, offs.data(), output_len * 8, hipMemcpyHostToDevice) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "copy offsets"));
    p->row_offsets_d = static_cast<const uint64_t *>(doffs.p);
    if (!win.empty()) {
        if (hipMalloc(&dwin.p, W * 4) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "hipMalloc window"));
        if (hipMemcpy(dwin.p, win.data(), W * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "copy window"));
        p->window_d = static_cast<const float *>(dwin.p);
    }
    const void *src = in; void *dst = rows;
    if (mem == QD_MEM_HOST) {
        if (hipMalloc(&din.p, n_in * 8) != hipSuccess || hipMalloc(&dout.p, output_len * W * 4) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "hipMalloc"));
        if (hipMemcpy(din.p, in, n_in * 8, hipMemcpyHostToDevice) != hipSuccess) return cleanup(fail(QD_ERR_HIP, "H2D"));
        src = din.p; dst = dout.p;
    }
    // rows are irregular: the per-sample kernel (no vector-alignment assumptions) reads them
    p->fn = p->fn_unaligned;
    p->d.n_samples = in_first + n_in;
    rc = launch_chain(p, src, in_first, n_in, 0, output_len, 0, dst, nullptr);
    if (rc == QD_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(QD_ERR_HIP, "sync failed");
    if (rc == QD_OK && mem == QD_MEM_HOST && hipMemcpy(rows, dout.p, output_len * W * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(QD_ERR_HIP, "D2H");
    return cleanup(rc);
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
    DevBuf dc, dout;
    HIPCHK(hipMalloc(&dc.p, n_cos * 8));
    HIPCHK(hipMemcpy(dc.p, cos_hz, n_cos * 8, hipMemcpyHostToDevice));
    float2 *o = reinterpret_cast<float2 *>(out);
    if (mem == QD_MEM_HOST) { HIPCHK(hipMalloc(&dout.p, n * 8)); o = static_cast<float2 *>(dout.p); }
    size_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_gen, dim3((uint32_t)blocks), dim3(256), 0, 0, static_cast<const int64_t *>(dc.p), (uint32_t)n_cos, sample_rate, first, n, o);
    HIPCHK(hipGetLastError());
    if (mem == QD_MEM_HOST) HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    else HIPCHK(hipDeviceSynchronize());
    return QD_OK;
}

}  // extern "C"
