// What does a ds_read_b128 cost next to v_mfma_f32_32x32x16_bf16?  The inner step of conv_halo.hip is [12 ds_read_b128 + 24 MFMA] per wave
// (8 waves per workgroup, 2 per SIMD) and its MFMA-pipe-busy figure follows  cycles/MFMA = 32 + ~38 x (reads per MFMA)  whatever is done about
// barriers, global-load placement or fragment prefetch.  This probe runs that step with NOTHING else, in hand-ordered inline asm:
//   V0  24 MFMA                                              (register operands only)
//   V1  24 MFMA + 12 reads into registers no MFMA uses, one read after every second MFMA
//   V2  the kernel's order: 4 reads, wait, 4 MFMA on them ... (single fragment set: a read overwrites operands of in-flight MFMAs)
//   V3  two fragment sets: the reads of iteration i+1 interleaved with the MFMAs of iteration i
//   V4  V1 with the 12 reads back to back at the top
//   V5  V1 with 6 reads           V6  V1 with 24 ds_read_b64 (same bytes)
//   V7  V3 with s_barrier per iteration
//   hipcc --offload-arch=gfx950 -O3 mfma_lds_mix.hip -o mfma_lds_mix && ./mfma_lds_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, int iters, int rnd) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = rnd ? (((unsigned)i * 2654435761u) ^ ((unsigned)i >> 3) * 40503u) & 0xbfffbfffu : 0x3f803f80u + (i & 3);   // (rnd: random mantissas and signs, exponents kept finite)
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;     // conflict-free 16-byte slots
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    i32x4 a[6], b[6], a2[6], b2[6], d[12];
    i32x2 e[24];
    for (int i = 0; i < 6; ++i) for (int q = 0; q < 4; ++q) { a[i][q] = rnd ? (int)((((unsigned)(threadIdx.x * 24 + i * 4 + q) * 2654435761u) & 0xbfffbfffu) | 0x30003000u) & 0xbfffbfff : 0x3f803f80 + i; b[i][q] = rnd ? (int)((((unsigned)(threadIdx.x * 31 + i * 4 + q + 7) * 2246822519u) & 0xbfffbfffu) | 0x30003000u) & 0xbfffbfff : 0x3e803e80 + q; a2[i][q] = a[i][q]; b2[i][q] = b[i][q]; }
    for (int i = 0; i < 12; ++i) for (int q = 0; q < 4; ++q) d[i][q] = 0;
    for (int i = 0; i < 24; ++i) for (int q = 0; q < 2; ++q) e[i][q] = 0;
    // 24 MFMAs on fragment sets (A, B): products (pa, pb) of the six-product pattern, accumulators cycling c0..c3
#define STEP4(A, B, pa, pb) MFMA(c0, A[pa], B[pb]); MFMA(c1, A[pa], B[pb + 3]); MFMA(c2, A[pa + 3], B[pb]); MFMA(c3, A[pa + 3], B[pb + 3]);
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
            STEP4(a, b, 2, 0) STEP4(a, b, 0, 2) STEP4(a, b, 1, 1) STEP4(a, b, 1, 0) STEP4(a, b, 0, 1) STEP4(a, b, 0, 0)
        } else if constexpr (V == 1 || V == 5) {
#define PAIR(i, A, B, pa, pb, x, y) MFMA(c0, A[pa], B[pb]); MFMA(c1, A[pa], B[pb + 3]); if (V == 1 || (i & 1) == 0) RD(d[x], addr, 0); MFMA(c2, A[pa + 3], B[pb]); MFMA(c3, A[pa + 3], B[pb + 3]); if (V == 1 || (i & 1) == 0) RD(d[y], addr, 512);
            PAIR(0, a, b, 2, 0, 0, 1) PAIR(1, a, b, 0, 2, 2, 3) PAIR(2, a, b, 1, 1, 4, 5) PAIR(3, a, b, 1, 0, 6, 7) PAIR(4, a, b, 0, 1, 8, 9) PAIR(5, a, b, 0, 0, 10, 11)
            WAIT(0);
        } else if constexpr (V == 6) {
#define QUAD(A, B, pa, pb, x) MFMA(c0, A[pa], B[pb]); RD64(e[x], addr, 0); MFMA(c1, A[pa], B[pb + 3]); RD64(e[x + 1], addr, 512); MFMA(c2, A[pa + 3], B[pb]); RD64(e[x + 2], addr, 1024); MFMA(c3, A[pa + 3], B[pb + 3]); RD64(e[x + 3], addr, 1536);
            QUAD(a, b, 2, 0, 0) QUAD(a, b, 0, 2, 4) QUAD(a, b, 1, 1, 8) QUAD(a, b, 1, 0, 12) QUAD(a, b, 0, 1, 16) QUAD(a, b, 0, 0, 20)
            WAIT(0);
        } else if constexpr (V == 4) {
            RD(d[0], addr, 0); RD(d[1], addr, 512); RD(d[2], addr, 1024); RD(d[3], addr, 1536); RD(d[4], addr, 2048); RD(d[5], addr, 2560);
            RD(d[6], addr, 0); RD(d[7], addr, 512); RD(d[8], addr, 1024); RD(d[9], addr, 1536); RD(d[10], addr, 2048); RD(d[11], addr, 2560);
            STEP4(a, b, 2, 0) STEP4(a, b, 0, 2) STEP4(a, b, 1, 1) STEP4(a, b, 1, 0) STEP4(a, b, 0, 1) STEP4(a, b, 0, 0)
            WAIT(0);
        } else if constexpr (V == 2) {
            // piece 2 of A with piece 0 of B, ...: each group reads what it needs right before (the compiler's schedule of the kernel)
            RD(a[2], addr, 0); RD(a[5], addr, 512); RD(b[0], addr, 1024); RD(b[3], addr, 1536); WAIT(0); STEP4(a, b, 2, 0)
            RD(a[0], addr, 0); RD(a[3], addr, 512); RD(b[2], addr, 1024); RD(b[5], addr, 1536); WAIT(0); STEP4(a, b, 0, 2)
            RD(a[1], addr, 0); RD(a[4], addr, 512); RD(b[1], addr, 1024); RD(b[4], addr, 1536); WAIT(0); STEP4(a, b, 1, 1)
            STEP4(a, b, 1, 0) STEP4(a, b, 0, 1) STEP4(a, b, 0, 0)
        } else {    // V == 3 / 7: two sets, reads of the other set between the MFMAs of this one
#define PAIR2(A, B, pa, pb, RA, x, y) MFMA(c0, A[pa], B[pb]); MFMA(c1, A[pa], B[pb + 3]); RD(RA[x], addr, 0); MFMA(c2, A[pa + 3], B[pb]); MFMA(c3, A[pa + 3], B[pb + 3]); RD(RA[y], addr, 512);
            PAIR2(a, b, 2, 0, a2, 0, 1) PAIR2(a, b, 0, 2, a2, 2, 3) PAIR2(a, b, 1, 1, a2, 4, 5) PAIR2(a, b, 1, 0, b2, 0, 1) PAIR2(a, b, 0, 1, b2, 2, 3) PAIR2(a, b, 0, 0, b2, 4, 5)
            WAIT(0);
            if constexpr (V == 7) asm volatile("s_barrier" ::: "memory");
            PAIR2(a2, b2, 2, 0, a, 0, 1) PAIR2(a2, b2, 0, 2, a, 2, 3) PAIR2(a2, b2, 1, 1, a, 4, 5) PAIR2(a2, b2, 1, 0, b, 0, 1) PAIR2(a2, b2, 0, 1, b, 2, 3) PAIR2(a2, b2, 0, 0, b, 4, 5)
            WAIT(0);
            if constexpr (V == 7) asm volatile("s_barrier" ::: "memory");
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int i = 0; i < 12; ++i) s += (float)d[i][0];
    for (int i = 0; i < 24; ++i) s += (float)e[i][0];
    for (int i = 0; i < 6; ++i) s += (float)(a[i][0] + b[i][1] + a2[i][2] + b2[i][3]);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static int g_rnd = 0;          // argv[2] = 1: random operand bits (the constant operands of the default run toggle few wires: lower power)
static int g_scale = 1;        // argv[1]: run length multiplier (x10 = 130 ms per launch: long enough for the power management to settle)
template <int V> void run(float* d, const char* name) {
    const int grid = 256, iters = 20000 * g_scale;
    const double per_iter = (V == 3 || V == 7) ? 48.0 : 24.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(512), 0, 0, d, 2000, g_rnd);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(512), 0, 0, d, (V == 3 || V == 7) ? iters / 2 : iters, g_rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mf = (double)grid * 8 * ((V == 3 || V == 7) ? iters / 2 : iters) * per_iter;
    const double flops = mf * (2.0 * 32 * 32 * 16);
    printf("%-78s %7.2f ms  %5.0f TFLOP/s bf16 (/6 = %3.0f)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 6);
}
int main(int argc, char** argv) {
    if (argc > 1) g_scale = atoi(argv[1]);
    if (argc > 2) g_rnd = atoi(argv[2]);
    printf("run length x%d, %s operands\n", g_scale, g_rnd ? "random" : "constant");
    float* d; hipMalloc(&d, 4096 * 512 * 4);
    run<0>(d, "V0 24 MFMA, register operands");
    run<1>(d, "V1 + 12 ds_read_b128 into unused registers, one per 2 MFMA");
    run<4>(d, "V4 + 12 ds_read_b128 into unused registers, back to back at the top");
    run<5>(d, "V5 + 6 ds_read_b128 into unused registers");
    run<6>(d, "V6 + 24 ds_read_b64 into unused registers");
    run<2>(d, "V2 kernel order: reads into the operands, single set, wait before use");
    run<3>(d, "V3 two operand sets: next step's reads between this step's MFMAs");
    run<7>(d, "V7 V3 + s_barrier per step");
    return 0;
}
