// Development probe: sustained rate of the FIR's packed multiply / add block (the asm statement of fir_tiled2_pk) per SIMD, at
// 1, 2 and 4 waves per SIMD, against independent packed ops and scalar f32 ops.
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/ubench_pk.hip -o scripts/ubench_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    extern __shared__ char smem[];
    v2f a0 = {seed, 0.f}, a1 = {0.f, seed};
    v2f x0 = {1.0f + threadIdx.x, 2.f}, x1 = {3.f, 4.f}, x2 = {5.f, 6.f}, x3 = {7.f, 8.f};
    v2f h01 = {seed, 0.5f}, h23 = {0.25f, 0.125f}, g01 = {0.3f, 0.7f}, g23 = {0.9f, 0.1f};
    v2f t0 = {0.f, 0.f}, t1 = t0, t2 = t0, t3 = t0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {          // the FIR block: 8 mul + 8 add, two dependent add chains
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
            } else if (MODE == 1) {   // 16 independent packed multiplies, no op_sel
                asm volatile("v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %4, %7\n\tv_pk_mul_f32 %3, %4, %8\n\t"
                             "v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %4, %7\n\tv_pk_mul_f32 %3, %4, %8\n\t"
                             "v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %4, %7\n\tv_pk_mul_f32 %3, %4, %8\n\t"
                             "v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %4, %7\n\tv_pk_mul_f32 %3, %4, %8"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x0), "v"(h01), "v"(h23), "v"(g01), "v"(g23));
            } else if (MODE == 2) {   // 16 independent packed multiplies WITH op_sel (one word for both halves)
                asm volatile("v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %2, %4, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %3, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %2, %4, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %3, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %2, %4, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %3, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                             "v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %2, %4, %7 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %3, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x0), "v"(h01), "v"(h23), "v"(g01), "v"(g23));
            } else if (MODE == 3) {   // 16 scalar f32 ops in 4 independent chains (mul, add alternating)
                float *p = reinterpret_cast<float *>(&t0);
                asm volatile("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %4, %8\n\t"
                             "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\t"
                             "v_mul_f32 %0, %0, %5\n\tv_mul_f32 %1, %1, %6\n\tv_mul_f32 %2, %2, %7\n\tv_mul_f32 %3, %3, %8\n\t"
                             "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8"
                             : "=&v"(t0.x), "=&v"(t1.x), "=&v"(t2.x), "=&v"(t3.x) : "v"(x0.x), "v"(h01.x), "v"(h23.x), "v"(g01.x), "v"(g23.x));
                (void)p;
            } else if (MODE == 5) {   // 16 v_fma_f64 in 4 independent chains
                double *q = reinterpret_cast<double *>(&t0);
                asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                             : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x0), "v"(h01));
                (void)q;
            } else if (MODE == 6) {   // 16 v_cvt_f32_f64
                asm volatile("v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\t"
                             "v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %3"
                             : "=&v"(t0.x), "=&v"(t1.x) : "v"(x0), "v"(h01));
            } else {                  // MODE 4: one serial chain of dependent packed adds
                asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\t"
                             "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1"
                             : "+v"(a0) : "v"(h01));
            }
        }
    }
    if (a0.x + a1.y + t0.x + t1.x + t2.x + t3.x == 12345.678f) out[threadIdx.x] = a0.x;
}
template <int MODE>
void run(const char *name, float *d) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD = workgroups of 256 threads per CU
        const int lds = wps == 1 ? 150 * 1024 : (wps == 2 ? 76 * 1024 : 38 * 1024);
        const int iters = 4000; float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), lds, 0, d, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        const double instr_per_simd = (double)iters * 16 * 16 * wps;           // VALU instructions issued on one SIMD
        printf("%-44s %d wave(s)/SIMD: %.3f ms  %.3f ns per instruction per SIMD\n", name, wps, best, best * 1e6 / instr_per_simd);
    }
}
int main() {
    float *d; hipMalloc(&d, 4096);
    run<0>("FIR block (8 pk_mul + 8 pk_add, 2 chains)", d);
    run<1>("independent v_pk_mul_f32", d);
    run<2>("independent v_pk_mul_f32 with op_sel", d);
    run<3>("scalar f32, 4 chains", d);
    run<4>("serial chain of v_pk_add_f32", d);
    run<5>("v_fma_f64, 4 chains", d);
    run<6>("v_cvt_f32_f64", d);
    return 0;
}
