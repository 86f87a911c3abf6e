// Development probe: does the f64 matrix pipe run beside the vector unit on gfx950?  Per wave: (a) 16 v_fma_f64, (b) 4
// v_mfma_f64_16x16x4_f64, (c) both interleaved.  If (c) takes max(a, b) the pipes overlap; if a + b they share the f64 datapath.
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/ubench_mfma64.hip -o scripts/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double a = seed + threadIdx.x, b = 0.5, c0 = 0, c1 = 1, c2 = 2, c3 = 3;
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 2) {
                asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                             "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            }
            if (MODE == 1 || MODE == 2) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            }
        }
    }
    if (c0 + c1 + c2 + c3 + acc0.x + acc1.y + acc2.z + acc3.w == 12345.678) out[threadIdx.x] = c0;
}
template <int MODE>
void run(const char *name, double *d, double valu_per_u, double mfma_per_u) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000; float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, iters, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double per_simd = (double)iters * 8 * 4;      // (unrolled bodies) x 4 waves per SIMD
    printf("%-40s %.3f ms   per body and wave: %.1f ns (%g v_fma_f64 + %g mfma)\n", name, best, best * 1e6 / per_simd, valu_per_u, mfma_per_u);
}
int main() {
    double *d; hipMalloc(&d, 4096);
    run<0>("8 v_fma_f64", d, 8, 0);
    run<1>("2 v_mfma_f64_16x16x4_f64", d, 0, 2);
    run<2>("8 v_fma_f64 + 2 v_mfma_f64_16x16x4_f64", d, 8, 2);
    return 0;
}
