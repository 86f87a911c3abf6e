// FETCH_SIZE calibration on gfx950 for the load widths the chain kernel uses: streams a known byte count with
// 16, 8 and 4 bytes per lane (coalesced, each byte read exactly once).
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_fetch scripts/ubench_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_cal -- build/ubench_fetch
// FETCH_SIZE (KiB) * 1024 / bytes is the factor to divide by (0.5 for 16 B/lane per MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class V>
__global__ __launch_bounds__(256) void k_read(const V *src, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 2ull << 30;
    void *src; uint32_t *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096);
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4 *)src, bytes / 16, out);
        hipLaunchKernelGGL(k_read<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2 *)src, bytes / 8, out);
        hipLaunchKernelGGL(k_read<uint32_t>, dim3(4096), dim3(256), 0, 0, (const uint32_t *)src, bytes / 4, out);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %zu\n", bytes);
    return 0;
}
