// Development probe: board power and shader clock per INSTRUCTION CLASS of the chain kernel, so that kernel changes can be
// ranked in joules instead of cycles (the cfg3' / cfg2 kernels run at the package power cap: time follows energy).
// Every class runs as 1024 workgroups of 256 threads (4 waves per SIMD, every CU), back to back for ~2.5 s, while this process
// samples the amdgpu hwmon files (power1_average / power1_input, freq1_input) — no child process, no rocm-smi.
//   J per wave-instruction = (P_class - P_floor) / (wave-instructions per second), P_floor = the "s_nop" class (clocks up,
//   waves resident, nothing switching in the vector units).
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/ubench_energy.hip -o scripts/ubench_energy
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

enum { NOP = 0, PK, F32, F64, CVT, LDS_LANE, LDS_BCAST, LDS_WRITE, FIR1, FIR2, HBM, LDS_LIN, LDS_LANE_ASM, LDS_LIN_ASM, NMODES };
static const char *kNames[NMODES] = {
    "s_nop (floor: waves resident, nothing issued to the VALU)", "v_pk_mul_f32 / v_pk_add_f32 (the FIR's block)", "v_mul_f32 / v_add_f32, 4 chains",
    "v_fma_f64, 4 chains", "v_cvt_f32_f64", "ds_read_b128 per lane, 272 B lane stride (FIR sample reads)", "ds_read_b128 broadcast (FIR tap reads)",
    "ds_write_b128 per lane (phase-1 park)", "FIR mix R=1: 8 packed VALU + 2 lane reads + 1 broadcast read", "FIR mix R=2: 16 packed VALU + 2 lane reads + 2 broadcast reads",
    "global_load_dwordx4 stream (16 B / lane, 1 GiB window)", "ds_read_b128 per lane, 16 B lane stride (one linear KiB per instruction)",
    "ds_read_b128 per lane, 272 B stride, result unused (asm; no accumulate)", "ds_read_b128 per lane, 16 B stride, result unused (asm)"};
// wave-instructions of the measured class per wave and outer iteration (16 inner steps)
static const double kPerIter[NMODES] = {256, 256, 256, 256, 256, 16, 16, 16, 8 * 11, 16 * 20, 4, 16, 16, 16};

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const f4 *stream, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 9 * 1024; i += 256) reinterpret_cast<float *>(smem)[i] = (float)i * seed;
    __syncthreads();
    const char *lane = smem + (threadIdx.x & 63) * 272;
    const char *uni = smem + 32 * 1024;
    const char *lin = smem + (threadIdx.x & 63) * 16;
    char *wl = smem + threadIdx.x * 16;
    v2f a0 = {seed, 0.f}, a1 = {0.f, seed};
    v2f x0 = {1.0f + threadIdx.x, 2.f}, x1 = {3.f, 4.f}, x2 = {5.f, 6.f}, x3 = {7.f, 8.f};
    v2f h01 = {seed, 0.5f}, h23 = {0.25f, 0.125f}, g01 = {0.3f, 0.7f}, g23 = {0.9f, 0.1f};
    v2f t0 = {0.f, 0.f}, t1 = t0, t2 = t0, t3 = t0;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = (u * 16 + (it & 7) * 256) & 8191;
            if (MODE == NOP) {
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                             "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
            } else if (MODE == PK || MODE == FIR1 || MODE == FIR2) {
                if (MODE == FIR1) {
                    if (u & 1) continue;                          // 8 blocks of (8 VALU + 3 reads) per 16 steps
                    const f4 s0 = *reinterpret_cast<const f4 *>(lane + off), s1 = *reinterpret_cast<const f4 *>(lane + off + 16);
                    const f4 hq = *reinterpret_cast<const f4 *>(uni + off);
                    x0 = v2f{s0.x, s0.y}; x1 = v2f{s0.z, s0.w}; x2 = v2f{s1.x, s1.y}; x3 = v2f{s1.z, s1.w}; h01 = v2f{hq.x, hq.y}; h23 = v2f{hq.z, hq.w};
                    asm volatile("v_pk_mul_f32 %1, %3, %7 op_sel_hi:[1,0]\n\t"
                                 "v_pk_mul_f32 %2, %4, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                                 "v_pk_add_f32 %0, %0, %1\n\t"
                                 "v_pk_mul_f32 %1, %5, %8 op_sel_hi:[1,0]\n\t"
                                 "v_pk_add_f32 %0, %0, %2\n\t"
                                 "v_pk_mul_f32 %2, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                                 "v_pk_add_f32 %0, %0, %1\n\t"
                                 "v_pk_add_f32 %0, %0, %2"
                                 : "+v"(a0), "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
                    continue;
                }
                if (MODE == FIR2) {
                    const f4 s0 = *reinterpret_cast<const f4 *>(lane + off), s1 = *reinterpret_cast<const f4 *>(lane + off + 16);
                    const f4 hq = *reinterpret_cast<const f4 *>(uni + off), gq = *reinterpret_cast<const f4 *>(uni + off + 4096);
                    x0 = v2f{s0.x, s0.y}; x1 = v2f{s0.z, s0.w}; x2 = v2f{s1.x, s1.y}; x3 = v2f{s1.z, s1.w};
                    h01 = v2f{hq.x, hq.y}; h23 = v2f{hq.z, hq.w}; g01 = v2f{gq.x, gq.y}; g23 = v2f{gq.z, gq.w};
                }
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
            } else if (MODE == F32) {
                asm volatile("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %4, %8\n\t"
                             "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\t"
                             "v_mul_f32 %0, %0, %5\n\tv_mul_f32 %1, %1, %6\n\tv_mul_f32 %2, %2, %7\n\tv_mul_f32 %3, %3, %8\n\t"
                             "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8"
                             : "=&v"(t0.x), "=&v"(t1.x), "=&v"(t2.x), "=&v"(t3.x) : "v"(x0.x), "v"(h01.x), "v"(h23.x), "v"(g01.x), "v"(g23.x));
            } else if (MODE == F64) {
                asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                             : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x0), "v"(h01));
            } else if (MODE == CVT) {
                asm volatile("v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3"
                             : "=&v"(t0.x), "=&v"(t1.x) : "v"(x0), "v"(h01));
            } else if (MODE == LDS_LANE) {
                acc += *reinterpret_cast<const f4 *>(lane + off);
            } else if (MODE == LDS_LIN) {
                acc += *reinterpret_cast<const f4 *>(lin + off);
            } else if (MODE == LDS_LANE_ASM || MODE == LDS_LIN_ASM) {
                f4 tmp;
                const uint32_t addr = (uint32_t)(uintptr_t)((MODE == LDS_LANE_ASM ? lane : lin) + off);
                asm volatile("ds_read_b128 %0, %1" : "=v"(tmp) : "v"(addr));
                if (u == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == LDS_BCAST) {
                acc += *reinterpret_cast<const f4 *>(uni + off);
            } else if (MODE == LDS_WRITE) {
                *reinterpret_cast<f4 *>(wl + ((u & 7) * 4096)) = acc;
                asm volatile("" ::: "memory");
            } else if (MODE == HBM) {
                if (u & 3) continue;                              // 4 loads per 16 steps
                const size_t idx = (gtid + ((size_t)(it * 4 + (u >> 2)) << 18)) & ((1u << 26) - 1);     // 1 GiB window of f4
                acc += stream[idx];
            }
        }
    }
    if (a0.x + a1.y + t0.x + t1.x + t2.x + t3.x + acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = a0.x + acc.x;
}

static std::string g_power, g_freq;
static void find_hwmon() {
    // the hwmon directory of the device this process computes on (PCI address from the HIP runtime); fall back to the first card
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
void run(float *d, const f4 *stream, double *floor_w) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = MODE == HBM ? 2000 : 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // ~2.5 s of back-to-back launches; power / clock sampled from the host every 50 ms after the first 0.7 s
    const auto t_start = std::chrono::steady_clock::now();
    double sum_w = 0, sum_mhz = 0, max_w = 0; int n_s = 0, launches = 0; float ms_total = 0;
    hipEventRecord(e0);
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double next_sample = 0.7;
    while (since() < 2.5) {
        for (int i = 0; i < 4; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 38 * 1024, 0, d, stream, iters, 1.0f); ++launches; }
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
    printf("%-78s %7.1f W (max %6.1f)  %6.0f MHz  %9.3e wave-instr/s  %8.1f pJ / wave-instr over floor  (%d launches, %.0f ms, %d samples)\n",
           kNames[MODE], w, max_w, mhz, rate, pj, launches, ms_total, n_s);
    fflush(stdout);
}

int main() {
    find_hwmon();
    printf("power file: %s\nclock file: %s\n", g_power.empty() ? "(none found)" : g_power.c_str(), g_freq.c_str());
    float *d; hipMalloc(&d, 4096);
    f4 *stream; hipMalloc(&stream, (size_t)1 << 30); hipMemset(stream, 0, (size_t)1 << 30);
    double floor_w = -1;
    run<NOP>(d, stream, &floor_w);
    run<PK>(d, stream, &floor_w);
    run<F32>(d, stream, &floor_w);
    run<F64>(d, stream, &floor_w);
    run<CVT>(d, stream, &floor_w);
    run<LDS_LANE>(d, stream, &floor_w);
    run<LDS_BCAST>(d, stream, &floor_w);
    run<LDS_WRITE>(d, stream, &floor_w);
    run<FIR1>(d, stream, &floor_w);
    run<FIR2>(d, stream, &floor_w);
    run<HBM>(d, stream, &floor_w);
    {   // the same stream out of memory with other allocation attributes (does the path through L2 / the Infinity Cache cost less?)
        f4 *unc = nullptr, *fine = nullptr;
        if (hipExtMallocWithFlags((void **)&unc, (size_t)1 << 30, hipDeviceMallocUncached) == hipSuccess && unc) {
            hipMemset(unc, 0, (size_t)1 << 30);
            printf("[hipDeviceMallocUncached] "); run<HBM>(d, unc, &floor_w);
            hipFree(unc);
        } else printf("hipDeviceMallocUncached: allocation failed\n");
        if (hipExtMallocWithFlags((void **)&fine, (size_t)1 << 30, hipDeviceMallocFinegrained) == hipSuccess && fine) {
            hipMemset(fine, 0, (size_t)1 << 30);
            printf("[hipDeviceMallocFinegrained] "); run<HBM>(d, fine, &floor_w);
            hipFree(fine);
        } else printf("hipDeviceMallocFinegrained: allocation failed\n");
    }
    run<LDS_LIN>(d, stream, &floor_w);
    run<LDS_LANE_ASM>(d, stream, &floor_w);
    run<LDS_LIN_ASM>(d, stream, &floor_w);
    return 0;
}
