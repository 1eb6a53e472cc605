// Sustained fp32 MFMA ceiling on this box: 4 waves/CU (1 per SIMD) x 4 independent accumulators, random operands,
// long enough (~tens of ms) for DVFS to settle.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            a = -a * 0.999f; b = b * 1.0001f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu, iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 1000, 0.5f, 0.25f);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * iters * 32.0 * (2.0 * 32 * 32 * 2);
            printf("waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  (=> %.0f MHz effective)\n", blocks_per_cu, ms, flops / ms / 1e9,
                   flops / ms / 1e9 / 157.3 * 2400);
        }
    }
    return 0;
}
