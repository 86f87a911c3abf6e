// Dependent-chain latency of the FIR's accumulate step on gfx950: packed (v_pk_mul/add_f32) vs scalar f32.
// hipcc --offload-arch=gfx950 -O3 -o build/ubench_chain scripts/ubench_chain.hip && build/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k_pk_chain(float *out, float a, float b, unsigned long long *cyc) {
    v2f acc = {a, b}, x = {b, a}; float h = a * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        v2f p, hh = {h, h};
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(hh));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(p));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc.x + acc.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_sc_chain(float *out, float a, float b, unsigned long long *cyc) {
    float ar = a, ai = b, xr = b, xi = a, h = a * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        float pr, pi;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(xr), "v"(h));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pi) : "v"(xi), "v"(h));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(ar) : "v"(ar), "v"(pr));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(ai) : "v"(ai), "v"(pi));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = ar + ai;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// two independent packed chains (register-tiled FIR, R = 2)
__global__ void k_pk_chain2(float *out, float a, float b, unsigned long long *cyc) {
    v2f acc0 = {a, b}, acc1 = {b, a}, x = {b, a}; float h = a * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        v2f p0, p1, hh = {h, h};
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(x), "v"(hh));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(x), "v"(hh));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc0) : "v"(acc0), "v"(p0));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc1) : "v"(acc1), "v"(p1));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc0.x + acc0.y + acc1.x + acc1.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(k_pk_chain, dim3(256), dim3(threads), 0, 0, out, 1.0f, 2.0f, cyc);
                if (k == 1) hipLaunchKernelGGL(k_sc_chain, dim3(256), dim3(threads), 0, 0, out, 1.0f, 2.0f, cyc);
                if (k == 2) hipLaunchKernelGGL(k_pk_chain2, dim3(256), dim3(threads), 0, 0, out, 1.0f, 2.0f, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const char *names[3] = {"pk 1 chain (re,im packed)", "scalar 2 chains (re | im)", "pk 2 chains"};
            printf("waves/SIMD %d  %-28s  %6.1f counter ticks per tap-step   kernel %.3f ms  (%.2f ns per step)\n", threads / 256, names[k],
                   (double)h / N, ms, ms * 1e6 / N);
        }
    }
    return 0;
}
