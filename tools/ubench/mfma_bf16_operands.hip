// Does the bf16 MFMA rate depend on operand register variety?  Same loop as mfma_bf16_peak.hip but the 24 MFMAs of an iteration
// use 3 x 2 different A fragments and 3 x 2 different B fragments in the product pattern of the split-operand kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int VARIETY>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[3][2], b[3][2];
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) {
        a[p][i][e] = (short)(0x3f80 + threadIdx.x + e + 7 * p + i);
        b[p][i][e] = (short)(0x3e80 - threadIdx.x + e + 5 * p + 3 * i);
    }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VARIETY ? a[PA[t]][i] : a[0][0], VARIETY ? b[PB[t]][j] : b[0][0], acc[i][j], 0, 0, 0);
        if (VARIETY == 2) {                           // also change the fragments every iteration (as freshly loaded ones would)
#pragma unroll
            for (int p = 0; p < 3; ++p) { a[p][0][it & 7] ^= 1; b[p][1][it & 7] ^= 1; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V> void run(float* d, const char* name) {
    const int grid = 256 * 3, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 1000);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * 24.0 * (2.0 * 32 * 32 * 16);
    printf("%-46s %.2f ms  %.0f TFLOP/s bf16 (/6 = %.0f)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 6);
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    run<0>(d, "same A/B registers for every MFMA");
    run<1>(d, "3x2 A and 3x2 B fragments, six-product pattern");
    run<2>(d, "... fragments modified every iteration");
    return 0;
}
