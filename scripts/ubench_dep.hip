// Development probe: what bounds ONE FIR wave per SIMD?  Cycles per instruction of (a) a dependent v_pk_add_f32 chain, (b) the FIR
// block's VALU pattern without its LDS reads, (c) the block as fir_pair issues it (2 per-lane ds_read_b128 + 1 broadcast read, three
// blocks of reads in flight), (d) two independent chains interleaved (two outputs per lane) — one wave per SIMD, one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_dep.hip -o scripts/ubench_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 9 * 1024; i += 256) reinterpret_cast<float *>(smem)[i] = (float)i * seed;
    __syncthreads();
    const char *lane = smem + (threadIdx.x & 63) * 272;
    const char *uni = smem + 32 * 1024;
    v2f a0 = {seed, 0.f}, a1 = {0.f, seed}, t0, t1, t2, t3;
    v2f x0 = {1.0f + threadIdx.x, 2.f}, x1 = {3.f, 4.f}, x2 = {5.f, 6.f}, x3 = {7.f, 8.f}, h01 = {seed, 0.5f}, h23 = {0.25f, 0.125f};
    f4 sa[3], sb[3], hh[3];
    for (int s = 0; s < 3; ++s) { sa[s] = *reinterpret_cast<const f4 *>(lane + 32 * s); sb[s] = *reinterpret_cast<const f4 *>(lane + 32 * s + 16); hh[s] = *reinterpret_cast<const f4 *>(uni + 16 * s); }
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int off = (u * 32 + (it & 7) * 512) & 8191;
            if (MODE == 0) {
                asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(x0));
            } else if (MODE == 1 || MODE == 2) {
                if (MODE == 2) {
                    const int s = u % 3;
                    x0 = v2f{sa[s].x, sa[s].y}; x1 = v2f{sa[s].z, sa[s].w}; x2 = v2f{sb[s].x, sb[s].y}; x3 = v2f{sb[s].z, sb[s].w}; h01 = v2f{hh[s].x, hh[s].y}; h23 = v2f{hh[s].z, hh[s].w};
                }
                asm volatile("v_pk_mul_f32 %1, %3, %7 op_sel_hi:[1,0]\n\t"
                             "v_pk_mul_f32 %2, %4, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_mul_f32 %1, %5, %8 op_sel_hi:[1,0]\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\t"
                             "v_pk_mul_f32 %2, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_add_f32 %0, %0, %2"
                             : "+v"(a0), "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
                if (MODE == 2) {
                    const int s = u % 3;
                    sa[s] = *reinterpret_cast<const f4 *>(lane + off); sb[s] = *reinterpret_cast<const f4 *>(lane + off + 16); hh[s] = *reinterpret_cast<const f4 *>(uni + off);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 3) {       // two outputs per lane: 16 VALU, chains interleaved
                asm volatile("v_pk_mul_f32 %2, %6, %10 op_sel_hi:[1,0]\n\t"
                             "v_pk_mul_f32 %3, %6, %11 op_sel_hi:[1,0]\n\t"
                             "v_pk_mul_f32 %4, %7, %10 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\t"
                             "v_pk_mul_f32 %5, %7, %11 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %1, %1, %3\n\t"
                             "v_pk_mul_f32 %2, %8, %10 op_sel_hi:[1,0]\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\t"
                             "v_pk_mul_f32 %3, %8, %11 op_sel_hi:[1,0]\n\t"
                             "v_pk_add_f32 %1, %1, %5\n\t"
                             "v_pk_mul_f32 %4, %9, %10 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\t"
                             "v_pk_mul_f32 %5, %9, %11 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_add_f32 %1, %1, %3\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\t"
                             "v_pk_add_f32 %1, %1, %5"
                             : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h01), "v"(h23));
            } else if (MODE == 4) {       // scalar form: two chains (re, im) of v_mul_f32 / v_add_f32, 4 taps
                asm volatile("v_mul_f32 %2, %6, %10\n\tv_mul_f32 %3, %7, %10\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                             "v_mul_f32 %4, %8, %11\n\tv_mul_f32 %5, %9, %11\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                             "v_mul_f32 %2, %6, %11\n\tv_mul_f32 %3, %7, %11\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                             "v_mul_f32 %4, %8, %10\n\tv_mul_f32 %5, %9, %10\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"
                             : "+v"(a0.x), "+v"(a0.y), "=&v"(t0.x), "=&v"(t0.y), "=&v"(t1.x), "=&v"(t1.y)
                             : "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "v"(h01.x), "v"(h01.y));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = c1 - c0;
    if (a0.x + a1.y + t0.x + t1.x == 12345.678f) out[threadIdx.x] = a0.x + a1.x + sa[0].x + sb[1].y + hh[2].z;
}

template <int MODE>
void run(const char *name, double valu_per_unit, int threads) {
    float *d; unsigned long long *c;
    hipMalloc(&d, 4096); hipMalloc(&c, 64 * 8);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 150 * 1024, 0, d, c, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: convert with the kernel's wall time instead — report per-unit time in ns from events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 150 * 1024, 0, d, c, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double units = (double)iters * 12;
    printf("%-64s %d waves/SIMD: %7.2f ns per unit, %6.2f ns per VALU instruction (%.0f per unit)\n", name, threads / 256, ms * 1e6 / units, ms * 1e6 / units / valu_per_unit, valu_per_unit);
    hipFree(d); hipFree(c);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("8 dependent v_pk_add_f32", 8, threads);
        run<1>("FIR block, VALU only (4 mul + 4 chained add)", 8, threads);
        run<2>("FIR block as issued (2 lane reads + 1 broadcast, 3 in flight)", 8, threads);
        run<3>("two outputs per lane, VALU only (16 packed ops)", 16, threads);
        run<4>("scalar form, two chains, 4 taps (16 f32 ops)", 16, threads);
    }
    return 0;
}
