// Optimizer step of the data-parallel training path as three HBM-bound launches over FLAT fp32 buffers
// (all parameters / gradients / Adam moments of a parameter group live in one allocation each):
//   adam_prepare   t += 1, bias-correction scalars, clear the clipping slots              (1 thread)
//   grad_absmax    inf-norm of a gradient segment (reference train.py:63-66, clip_grad_norm_(norm_type=inf))
//   adam_flat      clip coefficient + Adam(betas, eps) update (reference train.py:21-25, torch.optim.Adam)
// torch.optim.Adam + clip_grad_norm_ issue ~1 500 launches per step for the 311 parameter tensors of the path (most of
// them on 0-dim "step" tensors when the update has to be graph-capturable); here the work is 28 bytes per parameter
// (w, g, m, v read; w, m, v written) streamed once: 116 M parameters = 3.25 GB = ~0.6 ms at HBM rate.
#include "common.h"

namespace {

// state layout per group (floats): [0] t  [1] lr / (1 - b1^t)  [2] 1 / sqrt(1 - b2^t)  [3] lr  [4..8) |g|_inf slots (>= 0)
__global__ void adam_prepare_kernel(float* __restrict__ state_all, int ngroups, double beta1, double beta2) {
    if ((int)threadIdx.x >= ngroups || blockIdx.x != 0) return;
    float* state = state_all + (size_t)threadIdx.x * MRFA_ADAM_STATE_FLOATS;
    const float t = state[0] + 1.0f;
    state[0] = t;
    const double bc1 = 1.0 - pow(beta1, (double)t);
    const double bc2 = 1.0 - pow(beta2, (double)t);
    state[1] = (float)((double)state[3] / bc1);
    state[2] = (float)(1.0 / sqrt(bc2));
#pragma unroll
    for (int s = 0; s < MRFA_ADAM_CLIP_SLOTS; ++s) state[4 + s] = 0.0f;
}

__global__ void grad_absmax_kernel(const f32x4* __restrict__ g, long long n4, unsigned* __restrict__ slot) {
    float m = 0.0f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = g[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // non-negative floats order like their bit patterns: ONE integer atomic per workgroup (16 K same-address atomics
    // from one per wave serialised into 0.2 ms)
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.0f) atomicMax(slot, __float_as_uint(m));
    }
}

__global__ void adam_flat_kernel(f32x4* __restrict__ w, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                                 long long n4, const float* __restrict__ state, float omb1, float beta2, float omb2, float eps,
                                 float gscale, int clip_slot, float max_norm) {
    const float step_size = state[1];
    const float inv_bc2s = state[2];
    float coef = gscale;
    if (clip_slot >= 0) {
        // clip_grad_norm_: coef = min(1, max_norm / (|g|_inf + 1e-6)) on the (already averaged) gradient
        const float total = state[4 + clip_slot] * gscale;
        coef *= fminf(1.0f, max_norm / (total + 1e-6f));
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 gv = g[i], mv = m[i], vv = v[i], wv = w[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gv[k] * coef;
            const float mk = mv[k] + omb1 * (gk - mv[k]);                  // exp_avg.lerp_(grad, 1 - beta1)
            const float vk = beta2 * vv[k] + omb2 * gk * gk;                // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
            const float denom = sqrtf(vk) * inv_bc2s + eps;
            wv[k] -= step_size * (mk / denom);
            mv[k] = mk;
            vv[k] = vk;
        }
        w[i] = wv;
        m[i] = mv;
        v[i] = vv;
    }
}

}  // namespace

extern "C" int mrfa_adam_prepare(void* stream, float* state, int ngroups, double beta1, double beta2) {
    MRFA_CHECK_ARG(state != nullptr && ngroups >= 1 && ngroups <= 64, "adam_prepare: null state or ngroups %d not in [1, 64]", ngroups);
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, ngroups, beta1, beta2);
    MRFA_CHECK_LAUNCH("adam_prepare");
    return 0;
}

extern "C" int mrfa_grad_absmax(void* stream, const float* g, long long n, float* state, int clip_slot) {
    MRFA_CHECK_ARG(n % 4 == 0 && aligned16(g), "grad_absmax: n %% 4 != 0 or unaligned buffer");
    MRFA_CHECK_ARG(clip_slot >= 0 && clip_slot < MRFA_ADAM_CLIP_SLOTS, "grad_absmax: clip slot %d out of range", clip_slot);
    if (n == 0) return 0;
    int grid = stream_grid(n / 4, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(grad_absmax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(g), n / 4, reinterpret_cast<unsigned*>(state + 4 + clip_slot));
    MRFA_CHECK_LAUNCH("grad_absmax");
    return 0;
}

extern "C" int mrfa_adam_flat(void* stream, float* w, const float* g, float* m, float* v, long long n, const float* state, double beta1,
                              double beta2, float eps, float gscale, int clip_slot, float max_norm) {
    MRFA_CHECK_ARG(n % 4 == 0 && aligned16(w) && aligned16(g) && aligned16(m) && aligned16(v), "adam_flat: n %% 4 != 0 or unaligned buffer");
    MRFA_CHECK_ARG(clip_slot < MRFA_ADAM_CLIP_SLOTS, "adam_flat: clip slot %d out of range", clip_slot);
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_flat_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4*>(w),
                       reinterpret_cast<const f32x4*>(g), reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), n / 4, state,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, gscale, clip_slot, max_norm);   // as torch rounds them
    MRFA_CHECK_LAUNCH("adam_flat");
    return 0;
}
