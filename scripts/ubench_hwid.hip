// Development probe: which SIMD does wave k of a workgroup land on?  (HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
// build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_hwid.hip -o scripts/ubench_hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    extern __shared__ char smem[];
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    volatile char *s = smem; s[threadIdx.x] = 1;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = id; out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = xcc & 0xf; }
}
int main() {
    for (int nt : {256, 1024}) {
        const int wg = nt == 256 ? 1024 : 256, wpw = nt / 64, lds = nt == 256 ? 39424 : 158656;
        unsigned *d; hipMalloc(&d, 8 * wg * wpw);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(wg), dim3(nt), lds, 0, d, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * wg * wpw); hipMemcpy(h.data(), d, 8 * wg * wpw, hipMemcpyDeviceToHost);
        int hist[16][4] = {};
        for (int b = 0; b < wg; ++b) for (int w = 0; w < wpw; ++w) hist[w][(h[2 * (b * wpw + w)] >> 4) & 3]++;
        printf("threads %d: wave index -> SIMD histogram over %d workgroups\n", nt, wg);
        for (int w = 0; w < wpw; ++w) printf("  wave %2d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("  first workgroups (xcc se cu | simd of waves):\n");
        for (int b = 0; b < 12; ++b) { unsigned v = h[2 * b * wpw]; printf("   wg %d: xcc %u se %u cu %u |", b, h[2 * b * wpw + 1], (v >> 13) & 7, (v >> 8) & 15); for (int w = 0; w < wpw; ++w) printf(" %u", (h[2 * (b * wpw + w)] >> 4) & 3); printf("\n"); }
        // per CU: how many of its resident workgroups have wave 0 or wave 1 (the FIR waves of the 256-thread kernels) on each SIMD
        if (nt == 256) {
            int cu_load[8 * 8 * 16][4] = {}; int cu_wgs[8 * 8 * 16] = {};
            for (int b = 0; b < wg; ++b) {
                unsigned v0 = h[2 * (b * wpw)], v1 = h[2 * (b * wpw + 1)], xcc = h[2 * b * wpw + 1];
                int key = (int)((xcc * 8 + ((v0 >> 13) & 7)) * 16 + ((v0 >> 8) & 15));
                cu_load[key][(v0 >> 4) & 3]++; cu_load[key][(v1 >> 4) & 3]++; cu_wgs[key]++;
            }
            int pattern[5][5][5][5] = {}; int n_cu = 0;
            for (int kk = 0; kk < 8 * 8 * 16; ++kk) if (cu_wgs[kk]) { ++n_cu; int *l = cu_load[kk]; pattern[l[0] > 4 ? 4 : l[0]][l[1] > 4 ? 4 : l[1]][l[2] > 4 ? 4 : l[2]][l[3] > 4 ? 4 : l[3]]++; }
            printf("  per CU (%d CUs seen): FIR-wave count per SIMD (waves 0 and 1 of every resident workgroup) -> number of CUs\n", n_cu);
            for (int a = 0; a < 5; ++a) for (int b2 = 0; b2 < 5; ++b2) for (int c = 0; c < 5; ++c) for (int e = 0; e < 5; ++e)
                if (pattern[a][b2][c][e]) printf("    simd loads %d %d %d %d : %d CUs\n", a, b2, c, e, pattern[a][b2][c][e]);
        }
        hipFree(d);
    }
    return 0;
}
