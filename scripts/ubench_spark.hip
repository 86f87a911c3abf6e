// Development probe: what a per-WAVE tile walk can stream.  k_spark / k_spark2 (qd_chain.h) give every wave an 8 KiB tile whose rows it
// keeps in flight in registers; with every arithmetic phase ablated they still read only 3.4 TB/s (profiles/r04/spark_ablate.log).
// This bench isolates the load skeleton and varies ONE thing at a time:
//   pattern  0: row y of a tile = 64 lanes x 16 B contiguous (k_spark)      1: lane (g, xp) reads 16 B at g * 1 KiB + y * 128 + xp * 16 (k_spark2)
//   policy   0: plain global_load   1: buffer_load nt   2: buffer_load (default policy)
//   walk     0: chip-wide grid stride   1: each XCD group (blockIdx % 8) walks a contiguous eighth
//   rows     loads in flight per lane (8 = one tile; 16 = two tiles ahead)
//   consume  0: xor into a register   1: + LDS write (ds_write_b128 x rows)   2: + a 4 B/lane store stream of half the bytes read
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_spark.hip -o scripts/ubench_spark
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int PATTERN, int POLICY, int WALK, int ROWS, int CONSUME, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint8_t *src, size_t n_tiles, float *out, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t TILE = 8192;                      // bytes per tile (8 rows of 1 KiB)
    constexpr int TPL = ROWS / 8;                        // tiles in flight per wave
    uint64_t first, stride, end;
    if (WALK == 1) {
        const uint64_t n8 = (n_tiles + 7) / 8, lo = (uint64_t)(blockIdx.x & 7) * n8;
        first = lo + (uint64_t)(blockIdx.x >> 3) * 4 + wave; stride = (uint64_t)(gridDim.x >> 3) * 4; end = lo + n8 < n_tiles ? lo + n8 : n_tiles;
    } else { first = (uint64_t)blockIdx.x * 4 + wave; stride = (uint64_t)gridDim.x * 4; end = n_tiles; }
    const uint32_t voff = PATTERN == 0 ? lane * 16 : (lane >> 3) * 1024 + (lane & 7) * 16;
    const uint32_t rstep = PATTERN == 0 ? 1024 : 128;
    v4u pf[ROWS];
    auto load_tile = [&](uint64_t t, int slot) {
        const uint8_t *base = src + t * TILE;
        if (POLICY == 0) {
#pragma unroll
            for (int y = 0; y < 8; ++y) pf[slot * 8 + y] = *reinterpret_cast<const v4u *>(base + voff + y * rstep);
        } else {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 0xffffffffu, 0x00020000);
#pragma unroll
            for (int y = 0; y < 8; ++y) pf[slot * 8 + y] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(y * rstep), POLICY == 1 ? 2 : 0);
        }
    };
    uint64_t tile = first;
    if (tile >= end) return;
#pragma unroll
    for (int s = 0; s < TPL; ++s) { const uint64_t t = tile + s * stride; load_tile(t < end ? t : tile, s); }
    uint32_t acc = 0;
    v4u *lw = reinterpret_cast<v4u *>(smem) + wave * 512 + lane;
    int slot = 0;
    while (true) {
        const uint64_t tn = tile + TPL * stride;
#pragma unroll
        for (int s = 0; s < TPL; ++s) {
            if (s != slot) continue;                     // (TPL <= 2: the slot test folds after unrolling)
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const v4u v = pf[s * 8 + y];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
                if (CONSUME >= 1) lw[y * 64] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_tile(tn < end ? tn : tile, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CONSUME == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) out[tile * 1024 + q * 64 + lane] = (float)acc;       // 4 KiB of output per 8 KiB tile, 256 B per instruction
        } else if (CONSUME == 3 || CONSUME == 4) {                                            // the same bytes as 16-byte stores: 1 KiB per instruction
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = {(float)acc, 1.f, 2.f, 3.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 *dst = reinterpret_cast<f4 *>(out + tile * 1024) + q * 64 + lane;
                if (CONSUME == 4) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        } else if (CONSUME == 5) {                                                            // 4-byte stores, non-temporal
#pragma unroll
            for (int q = 0; q < 16; ++q) __builtin_nontemporal_store((float)acc, out + tile * 1024 + q * 64 + lane);
        } else if (CONSUME == 6) {                                                            // 16-byte stores in k_spark2's shape: 8 lanes x 16 B = 128 B per window, 8 windows 512 B apart
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = {(float)acc, 1.f, 2.f, 3.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) *(reinterpret_cast<f4 *>(out + tile * 1024 + (lane >> 3) * 128 + q * 32) + (lane & 7)) = v;
        }
        slot = (slot + 1) % TPL;
        tile += stride;
        if (tile >= end) break;
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int PATTERN, int POLICY, int WALK, int ROWS, int CONSUME, int WPS>
void run(const uint8_t *src, size_t bytes, float *out, uint32_t *sink, const char *what) {
    const size_t n_tiles = bytes / 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PATTERN, POLICY, WALK, ROWS, CONSUME, WPS>), dim3(256 * WPS), dim3(256), 4 * 8192, 0, src, n_tiles, out, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double moved = (double)bytes * (CONSUME >= 2 ? 1.5 : 1.0);
    printf("pattern %d policy %d walk %d rows %2d consume %d waves/SIMD %d  %-44s %7.3f ms  read %5.0f GB/s  total %5.0f GB/s\n", PATTERN, POLICY, WALK, ROWS, CONSUME, WPS, what,
           best, bytes / best / 1e6, moved / best / 1e6);
    fflush(stdout);
}

int main() {
    const size_t bytes = 8ull << 30;
    uint8_t *src; float *out; uint32_t *sink;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, bytes / 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4096);
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    run<0, 0, 0, 8, 0, 4>(src, bytes, out, sink, "contiguous rows, plain loads, chip-wide walk");
    run<0, 1, 0, 8, 0, 4>(src, bytes, out, sink, "... buffer loads nt");
    run<0, 2, 0, 8, 0, 4>(src, bytes, out, sink, "... buffer loads, default policy");
    run<0, 1, 1, 8, 0, 4>(src, bytes, out, sink, "... nt, XCD eighths");
    run<0, 0, 1, 8, 0, 4>(src, bytes, out, sink, "... plain, XCD eighths");
    run<1, 1, 1, 8, 0, 4>(src, bytes, out, sink, "128 B segments (k_spark2), nt, XCD eighths");
    run<1, 0, 1, 8, 0, 4>(src, bytes, out, sink, "128 B segments, plain, XCD eighths");
    run<0, 1, 1, 16, 0, 4>(src, bytes, out, sink, "contiguous, nt, XCD eighths, two tiles ahead");
    run<0, 1, 1, 8, 0, 8>(src, bytes, out, sink, "contiguous, nt, XCD eighths, 8 waves / SIMD");
    run<0, 1, 1, 8, 0, 2>(src, bytes, out, sink, "contiguous, nt, XCD eighths, 2 waves / SIMD");
    run<0, 1, 1, 8, 1, 4>(src, bytes, out, sink, "contiguous, nt, XCD eighths, + LDS writes");
    run<0, 1, 1, 8, 2, 4>(src, bytes, out, sink, "contiguous, nt, XCD eighths, + LDS + output stream");
    run<1, 1, 1, 8, 2, 4>(src, bytes, out, sink, "128 B segments, nt, XCD eighths, + LDS + output");
    run<0, 1, 1, 8, 3, 4>(src, bytes, out, sink, "... output as 16 B / lane stores");
    run<0, 1, 1, 8, 4, 4>(src, bytes, out, sink, "... 16 B / lane stores, nt");
    run<0, 1, 1, 8, 5, 4>(src, bytes, out, sink, "... 4 B / lane stores, nt");
    run<1, 1, 1, 8, 6, 4>(src, bytes, out, sink, "128 B segments in, 16 B stores in 128 B segments out");
    run<0, 1, 1, 8, 3, 8>(src, bytes, out, sink, "16 B stores, 8 waves / SIMD");
    run<0, 1, 1, 16, 3, 4>(src, bytes, out, sink, "16 B stores, two tiles ahead");
    run<0, 0, 0, 8, 2, 4>(src, bytes, out, sink, "contiguous, plain, chip-wide, + LDS + output");
    return 0;
}
