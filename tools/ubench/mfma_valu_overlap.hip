// Do a matrix-bound wave and a VALU-bound wave on the SAME SIMD overlap?  512-thread workgroups: waves 0-3 (one per SIMD)
// issue bf16 MFMAs back to back, waves 4-7 (their SIMD partners) run `valu_per_iter` dependent-free VALU ops per loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, unsigned seed) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3e80 - threadIdx.x + e); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        }
    } else if (mode & 2) {
        unsigned x[16];
        for (int e = 0; e < 16; ++e) x[e] = seed + threadIdx.x * 17 + e;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = (x[e] & 0xffff0000u) - (x[(e + 1) & 15] ^ 0x5bd1e995u);     // and, xor, sub: 3 VALU
        }
        for (int e = 0; e < 16; ++e) s += (float)x[e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 512 * 4);
    const int grid = 256, iters = 20000;
    const char* names[4] = {"", "MFMA waves only", "VALU waves only", "both on the same SIMDs"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, 1000, mode, 1u);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, mode, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double mfma = (double)grid * 4 * iters * 32.0, valu = (double)grid * 4 * iters * 8.0 * 16 * 3;
        printf("%-24s %.2f ms:  %.0f cycles/MFMA/SIMD@2.04GHz, %.2f cycles per VALU op per SIMD\n", names[mode], best,
               (mode & 1) ? best * 1e-3 * 2.04e9 / (mfma / (grid * 4)) : 0.0, (mode & 2) ? best * 1e-3 * 2.04e9 / (valu / (grid * 4)) : 0.0);
    }
    return 0;
}
