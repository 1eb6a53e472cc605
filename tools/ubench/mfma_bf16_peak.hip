// Sustained bf16 MFMA ceiling on this box (v_mfma_f32_32x32x16_bf16): 1 or 2 waves per SIMD x 4 independent accumulators,
// long enough (~tens of ms) for DVFS / the power cap to settle.   hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k(float* out, int iters, short a0, short b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(a0 + threadIdx.x + e); b[e] = (short)(b0 - threadIdx.x * 3 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            a[u & 7] ^= (short)0x0101;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu, iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 1000, (short)0x3f80, (short)0x3e80);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, (short)0x3f80, (short)0x3e80);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * iters * 32.0 * (2.0 * 32 * 32 * 16);
            printf("waves/SIMD=%d  %.2f ms  %.1f TFLOP/s bf16  (/6 = %.1f fp32-equivalent; => %.0f MHz effective)\n", blocks_per_cu, ms,
                   flops / ms / 1e9, flops / ms / 1e9 / 6, flops / ms / 1e9 / 2516.6 * 2400);
        }
    }
    return 0;
}
