// Development probe ("next round" 3a): can the f32 MFMA serve the EXACT-order FIR, and what does it cost in joules?
//
// Part A — rounding of ONE K = 1 step.  The reference accumulates  acc = fl(acc + fl(x * h))  (src/filter.rs:119: Rust never
// contracts), so an MFMA step D = A * B + C is usable in exact mode only if it rounds the product BEFORE the add.  For N random
// triples (a, b, c) — a and b wave-uniform, so the result does not depend on the operand layout; c per lane and per result
// register — the kernel runs one v_mfma_f32_4x4x1_16b_f32 and one v_mfma_f32_32x32x1_2b_f32 step and compares every result
// element with fmaf(a, b, c) (one rounding) and with __fadd_rn(__fmul_rn(a, b), c) (two roundings).  Only triples on which the
// two differ discriminate; their number is printed too.
//
// Part B — energy.  Same method as scripts/ubench_energy.hip: 1024 workgroups x 256 threads back to back for ~2.5 s per class,
// board power from the amdgpu hwmon files, pJ per wave-instruction over the s_nop floor, then per multiply-accumulate:
//   v_pk_mul_f32 + v_pk_add_f32   2 instructions = 128 separately rounded MACs (what the exact FIR issues)
//   v_pk_fma_f32                  1 instruction  = 128 fused MACs            (QD_MODE_FAST today)
//   v_mfma_f32_4x4x1_16b_f32      256 MACs   v_mfma_f32_16x16x1_4b_f32  1024   v_mfma_f32_32x32x1_2b_f32  2048
//   v_mfma_f32_16x16x4_f32        1024 MACs  v_mfma_f32_32x32x2_f32     2048
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/ubench_mfma32.hip -o scripts/ubench_mfma32
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <dirent.h>
#include <string>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));

// ---------------------------------------------------------------- part A
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// a float with a full random mantissa, exponent in [-e, e], random sign
__device__ __forceinline__ float rnd_float(uint32_t h, int e) {
    const uint32_t man = h & 0x7fffffu, sgn = (h >> 31) << 31;
    const int ex = 127 + (int)((h >> 23) % (uint32_t)(2 * e + 1)) - e;
    return __uint_as_float(sgn | ((uint32_t)ex << 23) | man);
}

// counters: [0] elements, [1] fma != mul+add (discriminating), [2] mfma == fma on those, [3] mfma == mul+add on those,
//           [4] mfma == fma overall, [5] mfma == mul+add overall, [6] neither
template <int SHAPE>      // 0: 4x4x1_16b (4 result registers), 1: 32x32x1_2b (32 result registers)
__global__ __launch_bounds__(256) void k_round(unsigned long long *cnt, uint32_t seed) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const float a = rnd_float(mix(wave * 2u + seed), 8), b = rnd_float(mix(wave * 2u + 1u + seed * 7919u), 8);
    constexpr int NR = SHAPE == 0 ? 4 : 32;
    float c[NR], d[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        // c near the product's magnitude so that the add rounds (and cancels) in interesting ways
        const float p = a * b;
        const uint32_t h = mix((wave * 64u + lane) * 37u + (uint32_t)r + seed * 104729u);
        c[r] = p * rnd_float(h, 3);
    }
    if constexpr (SHAPE == 0) {
        f4 cc = {c[0], c[1], c[2], c[3]};
        cc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, cc, 0, 0, 0);
        d[0] = cc.x; d[1] = cc.y; d[2] = cc.z; d[3] = cc.w;
    } else {
        f32v cc;
#pragma unroll
        for (int r = 0; r < 32; ++r) cc[r] = c[r];
        cc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, cc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 32; ++r) d[r] = cc[r];
    }
    unsigned long long n = 0, disc = 0, df = 0, dm = 0, of = 0, om = 0, nn = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float f = __builtin_fmaf(a, b, c[r]);
        const float m = __fadd_rn(__fmul_rn(a, b), c[r]);
        const uint32_t ub = __float_as_uint(d[r]), uf = __float_as_uint(f), um = __float_as_uint(m);
        ++n;
        if (uf != um) { ++disc; if (ub == uf) ++df; if (ub == um) ++dm; }
        if (ub == uf) ++of;
        if (ub == um) ++om;
        if (ub != uf && ub != um) ++nn;
    }
    atomicAdd(&cnt[0], n); atomicAdd(&cnt[1], disc); atomicAdd(&cnt[2], df); atomicAdd(&cnt[3], dm);
    atomicAdd(&cnt[4], of); atomicAdd(&cnt[5], om); atomicAdd(&cnt[6], nn);
}

// ---------------------------------------------------------------- part B
enum { NOP = 0, PKMA, PKFMA, M4, M16x1, M32x1, M16x4, M32x2, NMODES };
static const char *kNames[NMODES] = {
    "s_nop (floor)", "v_pk_mul_f32 + v_pk_add_f32 (exact FIR pair)", "v_pk_fma_f32 (QD_MODE_FAST)", "v_mfma_f32_4x4x1_16b_f32",
    "v_mfma_f32_16x16x1_4b_f32", "v_mfma_f32_32x32x1_2b_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32"};
// wave-instructions per wave and outer iteration, and multiply-accumulates per wave-instruction
static const double kPerIter[NMODES] = {256, 256, 256, 64, 32, 16, 32, 16};
static const double kMacs[NMODES] = {0, 64, 128, 256, 1024, 2048, 1024, 2048};       // PKMA: two instructions per 128 MACs

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    const float lanev = 1.0f + 0.001f * (float)(threadIdx.x & 63) * seed;
    v2f a0 = {seed, 0.5f * lanev}, a1 = {0.25f, seed * lanev}, a2 = a0, a3 = a1, t0 = {0.f, 0.f}, t1 = t0;
    v2f x0 = {lanev, 2.f * lanev}, h01 = {0.999f, 1.0001f};
    f4 c4[4];
    f16v c16[2];
    f32v c32;
    for (int i = 0; i < 4; ++i) c4[i] = f4{lanev, 0.5f, 0.25f * lanev, 2.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) c16[i][r] = lanev * (float)(r + 1);
    for (int r = 0; r < 32; ++r) c32[r] = lanev * (float)(r + 1);
    const float ma = 0.9999f * lanev, mb = 1.0001f / lanev;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == NOP) {
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                             "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
            } else if (MODE == PKMA) {       // 16 instructions: 8 mul + 8 add on four chains
                asm volatile("v_pk_mul_f32 %4, %6, %7\n\tv_pk_mul_f32 %5, %6, %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\t"
                             "v_pk_mul_f32 %4, %6, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %5, %6, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %5\n\t"
                             "v_pk_mul_f32 %4, %6, %7\n\tv_pk_mul_f32 %5, %6, %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\t"
                             "v_pk_mul_f32 %4, %6, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %5, %6, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(h01));
            } else if (MODE == PKFMA) {      // 16 fused instructions on four chains
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                             "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                             "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                             "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(h01));
            } else if (MODE == M4) {         // 4 per step, independent accumulators
#pragma unroll
                for (int i = 0; i < 4; ++i) c4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, mb, c4[i], 0, 0, 0);
            } else if (MODE == M16x1) {      // 2 per step
#pragma unroll
                for (int i = 0; i < 2; ++i) c16[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(ma, mb, c16[i], 0, 0, 0);
            } else if (MODE == M32x1) {      // 1 per step
                c32 = __builtin_amdgcn_mfma_f32_32x32x1f32(ma, mb, c32, 0, 0, 0);
            } else if (MODE == M16x4) {      // 2 per step (K = 4: 16x16 result, 4 registers)
#pragma unroll
                for (int i = 0; i < 2; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, c4[i], 0, 0, 0);
            } else if (MODE == M32x2) {      // 1 per step (K = 2: 32x32 result, 16 registers)
                c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, c16[0], 0, 0, 0);
            }
        }
    }
    float s = a0.x + a1.y + a2.x + a3.y + t0.x + t1.x + c4[0].x + c4[1].y + c4[2].z + c4[3].w + c16[0][3] + c16[1][7] + c32[5];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static std::string g_power, g_freq;
static void find_hwmon() {
    char bdf[64] = {0};
    std::vector<std::string> bases;
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, 0) == hipSuccess && bdf[0]) {
        for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
        bases.push_back(std::string("/sys/bus/pci/devices/") + bdf + "/hwmon");
    }
    for (int card = 0; card < 64; ++card) { char base[256]; snprintf(base, sizeof base, "/sys/class/drm/card%d/device/hwmon", card); bases.push_back(base); }
    for (const std::string &base : bases) {
        DIR *d = opendir(base.c_str());
        if (!d) continue;
        while (dirent *e = readdir(d)) {
            if (strncmp(e->d_name, "hwmon", 5)) continue;
            for (const char *pf : {"power1_average", "power1_input"}) {
                std::string p = base + "/" + e->d_name + "/" + pf;
                if (FILE *f = fopen(p.c_str(), "r")) { fclose(f); g_power = p; g_freq = base + "/" + e->d_name + "/freq1_input"; break; }
            }
            if (!g_power.empty()) break;
        }
        closedir(d);
        if (!g_power.empty()) break;
    }
}
static double read_num(const std::string &p) {
    if (p.empty()) return -1;
    FILE *f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

template <int MODE>
void run(float *d, double *floor_w) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t_start = std::chrono::steady_clock::now();
    double sum_w = 0, sum_mhz = 0, max_w = 0; int n_s = 0, launches = 0; float ms_total = 0;
    hipEventRecord(e0);
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double next_sample = 0.7;
    while (since() < 2.5) {
        for (int i = 0; i < 4; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, iters, 1.0f); ++launches; }
        hipStreamSynchronize(0);
        if (since() >= next_sample) {
            const double w = read_num(g_power) * 1e-6, mhz = read_num(g_freq) * 1e-6;
            if (w > 0) { sum_w += w; if (w > max_w) max_w = w; sum_mhz += mhz; ++n_s; }
            next_sample = since() + 0.05;
        }
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_total, e0, e1);
    const double wave_instr = (double)launches * 1024 * 4 * iters * kPerIter[MODE];
    const double rate = wave_instr / (ms_total * 1e-3);
    const double w = n_s ? sum_w / n_s : -1, mhz = n_s ? sum_mhz / n_s : -1;
    if (MODE == NOP) *floor_w = w;
    const double pj = (w > 0 && *floor_w > 0) ? (w - *floor_w) / rate * 1e12 : -1;
    const double macs = kMacs[MODE];
    printf("%-46s %7.1f W (max %6.1f) %6.0f MHz  %9.3e wave-instr/s  %9.1f pJ / wave-instr", kNames[MODE], w, max_w, mhz, rate, pj);
    if (macs > 0) printf("  %6.2f pJ / MAC  %8.3e MAC/s chip-wide", pj / macs, rate * macs);
    printf("  (%d launches, %.0f ms, %d samples)\n", launches, ms_total, n_s);
    fflush(stdout);
}

int main() {
    unsigned long long *cnt; hipMalloc(&cnt, 8 * sizeof *cnt);
    for (int shape = 0; shape < 2; ++shape) {
        unsigned long long h[8] = {0};
        hipMemset(cnt, 0, 8 * sizeof *cnt);
        const int blocks = shape == 0 ? 12288 : 2048;        // 4x4x1: 12288 x 4 waves x 256 elements = 1.26e7; 32x32x1: 2048 x 4 x 2048 = 1.68e7
        if (shape == 0) hipLaunchKernelGGL(k_round<0>, dim3(blocks), dim3(256), 0, 0, cnt, 12345u);
        else hipLaunchKernelGGL(k_round<1>, dim3(blocks), dim3(256), 0, 0, cnt, 54321u);
        hipDeviceSynchronize();
        hipMemcpy(h, cnt, sizeof h, hipMemcpyDeviceToHost);
        printf("rounding, %s: %llu result elements (a, b uniform per wave: %d products), %llu on which fmaf != fmul+fadd; of those MFMA == fmaf: %llu, == fmul_rn+fadd_rn: %llu; overall == fmaf: %llu, == mul+add: %llu, neither: %llu\n",
               shape == 0 ? "v_mfma_f32_4x4x1_16b_f32" : "v_mfma_f32_32x32x1_2b_f32", h[0], blocks * 4, h[1], h[2], h[3], h[4], h[5], h[6]);
        printf("  => one K = 1 MFMA step %s\n", h[2] == h[1] && h[1] > 0 ? "is FUSED (a single rounding, == fmaf): it cannot reproduce the reference's separately rounded multiply and add"
                                               : (h[3] == h[1] && h[1] > 0 ? "rounds the product BEFORE the add (== fmul_rn + fadd_rn): usable for the exact-order FIR" : "matches neither form on every element: see the counts"));
    }
    find_hwmon();
    printf("power file: %s\nclock file: %s\n", g_power.empty() ? "(none found)" : g_power.c_str(), g_freq.c_str());
    float *d; hipMalloc(&d, 4096);
    double floor_w = -1;
    run<NOP>(d, &floor_w);
    run<PKMA>(d, &floor_w);
    run<PKFMA>(d, &floor_w);
    run<M4>(d, &floor_w);
    run<M16x1>(d, &floor_w);
    run<M32x1>(d, &floor_w);
    run<M16x4>(d, &floor_w);
    run<M32x2>(d, &floor_w);
    return 0;
}
