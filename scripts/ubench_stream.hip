// Streaming-read ceiling for the chain kernel's access pattern on gfx950: persistent workgroups, each walking
// 33 KiB tiles (9 rows of 256 lanes x 16 B), the next tile's rows prefetched into registers while the current
// one is consumed.  Variants: consume = xor-reduce only | + LDS write/read + barriers (the chain's skeleton).
// hipcc --offload-arch=gfx950 -O3 -o build/ubench_stream scripts/ubench_stream.hip && build/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ROWS 9
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 4) void k_stream(const uint4 *src, size_t n_vec, size_t tile_vec, uint32_t *out) {
    extern __shared__ uint4 lds[];
    const size_t n_tiles = n_vec / tile_vec;
    uint4 pf[ROWS];
    size_t tile = blockIdx.x;
    if (tile < n_tiles) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) pf[i] = src[tile * tile_vec + (size_t)i * NT + threadIdx.x];
    }
    uint32_t acc = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        size_t nt = tile + gridDim.x; if (nt >= n_tiles) nt = tile;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            uint4 v = pf[i];
            pf[i] = src[nt * tile_vec + (size_t)i * NT + threadIdx.x];
            if (MODE == 0) acc ^= v.x ^ v.y ^ v.z ^ v.w;
            else lds[i * NT + threadIdx.x] = v;
        }
        if (MODE == 1) {
            __syncthreads();
            uint4 w = lds[(threadIdx.x * 7 + 3) % (ROWS * NT)];
            acc ^= w.x ^ w.y ^ w.z ^ w.w;
            __syncthreads();
            __syncthreads();
            __syncthreads();
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 1ull << 30, n_vec = bytes / 16;
    uint4 *src; uint32_t *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wg = 2; wg <= 8; wg *= 2) {
            const int NT = 256; const size_t tile_vec = ROWS * NT;
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((k_stream<0, 256>), dim3(256 * wg), dim3(NT), 0, 0, src, n_vec, tile_vec, out);
                else hipLaunchKernelGGL((k_stream<1, 256>), dim3(256 * wg), dim3(NT), ROWS * NT * 16, 0, src, n_vec, tile_vec, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
            }
            printf("mode %d (%s)  %d WG/CU: %.3f ms  %.0f GB/s\n", mode, mode ? "LDS write + 4 barriers" : "registers only", wg, best, bytes / best / 1e6);
        }
    return 0;
}
