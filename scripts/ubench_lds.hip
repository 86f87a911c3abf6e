// Development probe: LDS read throughput per CU on gfx950 for the FIR's access patterns: per-lane ds_read_b128 at a lane stride
// of 144 B (the padded register-tiled layout), at 264 B (the lane-per-output layout, D = 32 + 1 pad pair), broadcast
// ds_read_b128 (tap reads), ds_read_b64, and the FIR's 2 : 1 mix.  4 workgroups of 256 threads per CU.
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/ubench_lds.hip -o scripts/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 9 * 1024; i += 256) reinterpret_cast<float *>(smem)[i] = (float)i;
    __syncthreads();
    const char *lane = smem + (threadIdx.x & 63) * stride_bytes;       // every wave reads the same 64-lane footprint
    const char *uni = smem + 32 * 1024;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = (u * 16 + (it & 7) * 256) & 8191;
            if (MODE == 0) { acc += *reinterpret_cast<const f4 *>(lane + off); }
            else if (MODE == 1) { acc += *reinterpret_cast<const f4 *>(uni + off); }
            else if (MODE == 2) { const f2 v = *reinterpret_cast<const f2 *>(lane + off); acc.x += v.x; acc.y += v.y; }
            else { acc += *reinterpret_cast<const f4 *>(lane + off); if ((u % 3) == 2) acc += *reinterpret_cast<const f4 *>(uni + off); }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}
template <int MODE>
void run(const char *name, float *d, int stride_bytes, double reads_per_u, double bytes_per_lane_read) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000; float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 38 * 1024, 0, d, iters, stride_bytes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double wave_reads_per_cu = (double)iters * 16 * reads_per_u * 16;         // 16 waves per CU
    printf("%-52s stride %4d B: %.3f ms  %.2f ns per wave-read per CU  = %.0f lane-bytes / ns / CU\n", name, stride_bytes, best,
           best * 1e6 / wave_reads_per_cu, 64 * bytes_per_lane_read * wave_reads_per_cu / (best * 1e6));
}
int main() {
    float *d; hipMalloc(&d, 4096);
    run<0>("ds_read_b128 per lane", d, 144, 1, 16);
    run<0>("ds_read_b128 per lane", d, 272, 1, 16);
    run<0>("ds_read_b128 per lane (contiguous)", d, 16, 1, 16);
    run<1>("ds_read_b128 broadcast (one address per wave)", d, 0, 1, 16);
    run<2>("ds_read_b64 per lane", d, 72, 1, 8);
    run<2>("ds_read_b64 per lane", d, 264, 1, 8);
    run<3>("mix: 3 per-lane b128 : 1 broadcast b128 (approx.)", d, 144, 1.3125, 16);
    return 0;
}
