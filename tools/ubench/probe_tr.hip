// Probe of ds_read_b64_tr_b16 (gfx950): every 16-bit LDS element holds its own index; each lane reads 4 elements with the transpose
// read at an address of the [4 k][16 col] block pattern (lane l of a 16-lane group: row l / 4, columns 4 (l % 4) ..), and the host
// prints which (row, col) every lane received.  Row pitch PITCH elements.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
constexpr int PITCH = 16;
__global__ void probe(unsigned short* out) {
    __shared__ unsigned short lds[64 * PITCH];
    for (int i = threadIdx.x; i < 64 * PITCH; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // group g reads rows 4g .. 4g+3
    const unsigned short* a = lds + (4 * grp + l / 4) * PITCH + 4 * (l % 4);
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(a));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = r[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 64; ++t) {
        printf("lane %2d:", t);
        for (int e = 0; e < 4; ++e) printf(" (r%2d,c%2d)", h[t * 4 + e] / PITCH, h[t * 4 + e] % PITCH);
        printf("\n");
    }
    return 0;
}
