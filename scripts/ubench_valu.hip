// Micro-benchmark: per-SIMD issue cost of the VALU ops the chain kernel leans on (gfx950).
// Each kernel runs ITER iterations of 8 independent chains of one op; WAVES waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 4096;

template <int OP> __global__ void k(double *out, double a, double b) {
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3, f4 = (float)x4, f5 = (float)x5, f6 = (float)x6, f7 = (float)x7;
    float fb = (float)b;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6};
    v2 pb = {fb, fb};
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
                       x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a); }
        if (OP == 1) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b; x4 = x4 * b; x5 = x5 * b; x6 = x6 * b; x7 = x7 * b; }
        if (OP == 2) { x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b; }
        if (OP == 3) { f0 = (float)x0; x0 = x0 + f0; f1 = (float)x1; x1 = x1 + f1; f2 = (float)x2; x2 = x2 + f2; f3 = (float)x3; x3 = x3 + f3; }   // 4 cvt + 4 cvt back + 4 add
        if (OP == 4) { f0 = f0 * fb; f1 = f1 * fb; f2 = f2 * fb; f3 = f3 * fb; f4 = f4 * fb; f5 = f5 * fb; f6 = f6 * fb; f7 = f7 * fb; }
        if (OP == 5) { p0 = p0 * pb; p1 = p1 * pb; p2 = p2 * pb; p3 = p3 * pb; p4 = p4 * pb; p5 = p5 * pb; p6 = p6 * pb; p7 = p7 * pb; }
        if (OP == 6) { p0 = p0 + pb; p1 = p1 + pb; p2 = p2 + pb; p3 = p3 + pb; p4 = p4 + pb; p5 = p5 + pb; p6 = p6 + pb; p7 = p7 + pb; }
        if (OP == 7) { f0 = f0 + fb; f1 = f1 + fb; f2 = f2 + fb; f3 = f3 + fb; f4 = f4 + fb; f5 = f5 + fb; f6 = f6 + fb; f7 = f7 + fb; }
        if (OP == 8) { x0 = sqrt(x0); x1 = sqrt(x1); x2 = sqrt(x2); x3 = sqrt(x3); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP> void run(const char *name, int ops_per_iter, int waves_per_simd, double *d) {
    int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: waves_per_simd waves * ITER * ops wave-instructions
    double instr = (double)waves_per_simd * ITER * ops_per_iter;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main() {
    double *d; CHK(hipMalloc(&d, 256 * 8 * 256 * sizeof(double)));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", 8, w, d); run<1>("v_mul_f64", 8, w, d); run<2>("v_add_f64", 8, w, d);
        run<3>("cvt f32<->f64 + add", 12, w, d);
        run<4>("v_mul_f32", 8, w, d); run<7>("v_add_f32", 8, w, d); run<5>("v_pk_mul_f32", 8, w, d); run<6>("v_pk_add_f32", 8, w, d);
        run<8>("sqrt(f64) (IEEE)", 4, w, d);
    }
    return 0;
}
