// K14-K17: the prior-motion stage's small dense ops as fused HIP kernels (round 1 ran them as ~100 ATen launches per step inside
// "torch islands").  Everything here is ALU / latency bound on tensors of at most (B, 11, 64, 64, 2) floats; the point of the
// fusion is launch count and intermediate traffic (the reference materialises (B,K,h,w,2) grids with .repeat(), calls a batched
// LAPACK-style inverse for 2x2 matrices, and runs softmax / weighted sums as separate passes).
//
//   kp_gaussian      util.py:59-87 (kp2gaussian) [+ pos_embedding, raft.py:177-178]                      -> NHWC slice
//   prior_motion     dense_motion.py:36-46 (heat-map differences), 48-76 (sparse motions, closed-form 2x2 inverse, background
//                    affine), 78-85 (the K+1 warps of the 1/4-scale source, grid_sample align_corners=False) and the channel
//                    interleave of :118 -> hourglass input, motions, sparse_deformed
//   softmax_combine  dense_motion.py:129-136 (softmax over the K+1 motions, mask-weighted deformation)
//   kp_head          kp_detector.py:90-120 (spatial softmax at temperature T, soft-argmax, heat-map-weighted Jacobian pooling)
//
// Reductions: wave shuffles (wave_sum) + one LDS combine per workgroup; one workgroup owns a whole (sample, keypoint) row, so the
// parameter-sized gradients are written by exactly one thread each (deterministic order inside the kernel).
#include "common.h"

namespace {

constexpr int NT = 256;

template <int NV>
__device__ __forceinline__ void block_reduce(float (&v)[NV], float* lds /* [NV][4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = wave_sum(v[i]);
        if (lane == 0) lds[i * 4 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = lds[i * 4] + lds[i * 4 + 1] + lds[i * 4 + 2] + lds[i * 4 + 3];
    __syncthreads();
}

__device__ __forceinline__ float block_max(float v, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    v = fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
    __syncthreads();
    return v;
}

// make_coordinate_grid (util.py:90-108): x = 2 j / (w - 1) - 1
__device__ __forceinline__ float gcoord(int j, int n) { return 2.f * ((float)j / (float)(n - 1)) - 1.f; }

// ------------------------------------------------------------------------------------------------ kp_gaussian
__global__ __launch_bounds__(NT) void kp_gaussian_fwd_kernel(const float* __restrict__ kp, const float* __restrict__ pos, int B, int K, int H,
                                                             int W, float inv_var, float* __restrict__ out, int ldo) {
    const long long total = (long long)B * H * W * K;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int k = (int)(i % K);
        const long long pix = i / K;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
        const float dx = gcoord(x, W) - kp[((size_t)b * K + k) * 2], dy = gcoord(y, H) - kp[((size_t)b * K + k) * 2 + 1];
        float v = expf(-0.5f * (dx * dx + dy * dy) * inv_var);
        if (pos) v += pos[((size_t)k * H + y) * W + x];
        out[(size_t)pix * ldo + k] = v;
    }
}

// one workgroup per (b, k): dkp[b,k,:] += sum_pixels dout * G * (g - kp) / var;  dpos[k,y,x] += sum_b dout (atomic over b)
__global__ __launch_bounds__(NT) void kp_gaussian_bwd_kernel(const float* __restrict__ kp, int B, int K, int H, int W, float inv_var,
                                                             const float* __restrict__ dout, int lddo, float* __restrict__ dkp,
                                                             float* __restrict__ dpos) {
    __shared__ float lds[2 * 4];
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const float kx = kp[((size_t)b * K + k) * 2], ky = kp[((size_t)b * K + k) * 2 + 1];
    float acc[2] = {0.f, 0.f};
    for (int p = threadIdx.x; p < H * W; p += NT) {
        const int x = p % W, y = p / W;
        const float d = dout[((size_t)b * H * W + p) * lddo + k];
        const float dx = gcoord(x, W) - kx, dy = gcoord(y, H) - ky;
        const float g = d * expf(-0.5f * (dx * dx + dy * dy) * inv_var) * inv_var;
        acc[0] += g * dx;
        acc[1] += g * dy;
        if (dpos) atomicAdd(dpos + (size_t)k * H * W + p, d);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0 && dkp) {
        atomicAdd(dkp + ((size_t)b * K + k) * 2, acc[0]);
        atomicAdd(dkp + ((size_t)b * K + k) * 2 + 1, acc[1]);
    }
}

// ------------------------------------------------------------------------------------------------ prior_motion
struct Affine {      // motion_k(z) = J (z - kd) + ks   (k >= 1);  J = js * inv(jd) or the identity
    float j00, j01, j10, j11, kdx, kdy, ksx, ksy;
};

__device__ __forceinline__ Affine load_affine(const mrfa_prior_params& p, int b, int k /* >= 1 */) {
    Affine a;
    const size_t o = (size_t)b * p.K + (k - 1);
    a.kdx = p.kd[o * 2]; a.kdy = p.kd[o * 2 + 1];
    a.ksx = p.ks[o * 2]; a.ksy = p.ks[o * 2 + 1];
    if (p.jd) {
        const float a0 = p.jd[o * 4], a1 = p.jd[o * 4 + 1], a2 = p.jd[o * 4 + 2], a3 = p.jd[o * 4 + 3];
        const float idet = 1.f / (a0 * a3 - a1 * a2);
        const float i00 = a3 * idet, i01 = -a1 * idet, i10 = -a2 * idet, i11 = a0 * idet;
        const float s0 = p.js[o * 4], s1 = p.js[o * 4 + 1], s2 = p.js[o * 4 + 2], s3 = p.js[o * 4 + 3];
        a.j00 = s0 * i00 + s1 * i10; a.j01 = s0 * i01 + s1 * i11;
        a.j10 = s2 * i00 + s3 * i10; a.j11 = s2 * i01 + s3 * i11;
    } else {
        a.j00 = 1.f; a.j01 = 0.f; a.j10 = 0.f; a.j11 = 1.f;
    }
    return a;
}

// one thread per (b, k, y, x): heat-map difference, sparse motion, bilinear warp of the 1/4-scale source (align_corners=False,
// zeros outside), all written where their consumers read them
__global__ __launch_bounds__(NT) void prior_motion_fwd_kernel(const mrfa_prior_params p) {
    const int K1 = p.K + 1, HW = p.H * p.W, C = p.C;
    const long long total = (long long)p.B * K1 * HW;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int pix = (int)(i % HW);
        const int k = (int)((i / HW) % K1), b = (int)(i / ((long long)HW * K1));
        const int x = pix % p.W, y = pix / p.W;
        const float gx = gcoord(x, p.W), gy = gcoord(y, p.H);
        float mx, my, heat = 0.f;
        if (k == 0) {
            mx = gx; my = gy;
            if (p.bg) {                                           // dense_motion.py:69-73
                const float* m = p.bg + (size_t)b * 9;
                const float hx = m[0] * gx + m[1] * gy + m[2], hy = m[3] * gx + m[4] * gy + m[5], hz = m[6] * gx + m[7] * gy + m[8];
                mx = hx / hz; my = hy / hz;
            }
        } else {
            const Affine a = load_affine(p, b, k);
            const float zx = gx - a.kdx, zy = gy - a.kdy;
            mx = a.j00 * zx + a.j01 * zy + a.ksx;
            my = a.j10 * zx + a.j11 * zy + a.ksy;
            const float sx = gx - a.ksx, sy = gy - a.ksy;
            heat = expf(-0.5f * (zx * zx + zy * zy) * p.inv_var) - expf(-0.5f * (sx * sx + sy * sy) * p.inv_var);
        }
        const size_t mrow = ((size_t)(b * K1 + k) * HW + pix);
        p.motions[mrow * p.ldm] = mx;
        p.motions[mrow * p.ldm + 1] = my;
        float* dst = p.inp + ((size_t)b * HW + pix) * p.ldi + (size_t)k * (C + 1);
        dst[0] = heat;
        const float ix = ((mx + 1.f) * (float)p.W - 1.f) * 0.5f, iy = ((my + 1.f) * (float)p.H - 1.f) * 0.5f;
        const bool inside = ix > -1.f && iy > -1.f && ix < (float)p.W && iy < (float)p.H;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float fx = ix - fx0, fy = iy - fy0;
        const bool x0ok = inside && x0 >= 0, x1ok = inside && x0 + 1 < p.W, y0ok = inside && y0 >= 0, y1ok = inside && y0 + 1 < p.H;
        const float* src = p.src + (size_t)b * HW * p.lds;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (y0ok && x0ok) v += (1.f - fx) * (1.f - fy) * src[((size_t)y0 * p.W + x0) * p.lds + c];
            if (y0ok && x1ok) v += fx * (1.f - fy) * src[((size_t)y0 * p.W + x0 + 1) * p.lds + c];
            if (y1ok && x0ok) v += (1.f - fx) * fy * src[((size_t)(y0 + 1) * p.W + x0) * p.lds + c];
            if (y1ok && x1ok) v += fx * fy * src[((size_t)(y0 + 1) * p.W + x0 + 1) * p.lds + c];
            dst[1 + c] = v;
            if (p.sparse) p.sparse[(((size_t)(b * K1 + k) * C + c) * HW) + pix] = v;
        }
    }
}

// one workgroup per (b, k): the gradients of every consumer of motion_k / heat_k / warp_k reduced over the pixels, then one thread
// pushes the 2 + 4 sums through the affine map (closed-form inverse) into d kd, d ks, d jd, d js (k >= 1) or d bg (k == 0)
__global__ __launch_bounds__(NT) void prior_motion_bwd_kernel(const mrfa_prior_params p) {
    __shared__ float lds[10 * 4];
    const int K1 = p.K + 1, HW = p.H * p.W, C = p.C;
    const int b = blockIdx.x / K1, k = blockIdx.x % K1;
    Affine a = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
    if (k > 0) a = load_affine(p, b, k);
    const float* m = p.bg ? p.bg + (size_t)b * 9 : nullptr;
    // k >= 1: [0,1] sum dm, [2..5] sum dm z^T, [6,7] sum dheat Gd (g - kd) / var, [8,9] sum dheat Gs (g - ks) / var;  k == 0: [0..8] d bg
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    const float* src = p.src + (size_t)b * HW * p.lds;
    for (int pix = threadIdx.x; pix < HW; pix += NT) {
        const int x = pix % p.W, y = pix / p.W;
        const float gx = gcoord(x, p.W), gy = gcoord(y, p.H);
        const size_t mrow = ((size_t)(b * K1 + k) * HW + pix);
        const float mx = p.motions[mrow * p.ldm], my = p.motions[mrow * p.ldm + 1];
        float dmx = p.dmotions ? p.dmotions[mrow * p.ldm] : 0.f, dmy = p.dmotions ? p.dmotions[mrow * p.ldm + 1] : 0.f;
        const float* dd = p.dinp + ((size_t)b * HW + pix) * p.lddi + (size_t)k * (C + 1);
        // d(warp) / d(sampling position), as grid_sample_bwd (align_corners=False: ix = ((mx + 1) W - 1) / 2)
        const float ix = ((mx + 1.f) * (float)p.W - 1.f) * 0.5f, iy = ((my + 1.f) * (float)p.H - 1.f) * 0.5f;
        if (ix > -1.f && iy > -1.f && ix < (float)p.W && iy < (float)p.H) {
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float fx = ix - fx0, fy = iy - fy0;
            const bool x0ok = x0 >= 0, x1ok = x0 + 1 < p.W, y0ok = y0 >= 0, y1ok = y0 + 1 < p.H;
            float gix = 0.f, giy = 0.f;
            for (int c = 0; c < C; ++c) {
                float g = dd[1 + c];
                if (p.dsparse) g += p.dsparse[(((size_t)(b * K1 + k) * C + c) * HW) + pix];
                const float v00 = (y0ok && x0ok) ? src[((size_t)y0 * p.W + x0) * p.lds + c] : 0.f;
                const float v01 = (y0ok && x1ok) ? src[((size_t)y0 * p.W + x0 + 1) * p.lds + c] : 0.f;
                const float v10 = (y1ok && x0ok) ? src[((size_t)(y0 + 1) * p.W + x0) * p.lds + c] : 0.f;
                const float v11 = (y1ok && x1ok) ? src[((size_t)(y0 + 1) * p.W + x0 + 1) * p.lds + c] : 0.f;
                gix += g * ((v01 - v00) * (1.f - fy) + (v11 - v10) * fy);
                giy += g * ((v10 - v00) * (1.f - fx) + (v11 - v01) * fx);
            }
            dmx += gix * 0.5f * (float)p.W;
            dmy += giy * 0.5f * (float)p.H;
        }
        if (k == 0) {
            if (m) {
                const float hx = m[0] * gx + m[1] * gy + m[2], hy = m[3] * gx + m[4] * gy + m[5], hz = m[6] * gx + m[7] * gy + m[8];
                const float dhx = dmx / hz, dhy = dmy / hz, dhz = -(dmx * hx + dmy * hy) / (hz * hz);
                acc[0] += dhx * gx; acc[1] += dhx * gy; acc[2] += dhx;
                acc[3] += dhy * gx; acc[4] += dhy * gy; acc[5] += dhy;
                acc[6] += dhz * gx; acc[7] += dhz * gy; acc[8] += dhz;
            }
        } else {
            const float zx = gx - a.kdx, zy = gy - a.kdy, sx = gx - a.ksx, sy = gy - a.ksy;
            acc[0] += dmx; acc[1] += dmy;
            acc[2] += dmx * zx; acc[3] += dmx * zy; acc[4] += dmy * zx; acc[5] += dmy * zy;
            const float dh = dd[0];
            const float gd = dh * expf(-0.5f * (zx * zx + zy * zy) * p.inv_var) * p.inv_var;
            const float gs = dh * expf(-0.5f * (sx * sx + sy * sy) * p.inv_var) * p.inv_var;
            acc[6] += gd * zx; acc[7] += gd * zy;
            acc[8] += gs * sx; acc[9] += gs * sy;
        }
    }
    block_reduce<10>(acc, lds);
    if (threadIdx.x != 0) return;
    if (k == 0) {
        if (m && p.dbg)
            for (int i = 0; i < 9; ++i) atomicAdd(p.dbg + (size_t)b * 9 + i, acc[i]);
        return;
    }
    const size_t o = (size_t)b * p.K + (k - 1);
    // motion = J z + ks, z = g - kd  ->  d ks += sum dm - (heat: -Gs term),  d kd += -J^T sum dm + (heat: Gd term)
    if (p.dks) {
        atomicAdd(p.dks + o * 2, acc[0] - acc[8]);
        atomicAdd(p.dks + o * 2 + 1, acc[1] - acc[9]);
    }
    if (p.dkd) {
        atomicAdd(p.dkd + o * 2, -(a.j00 * acc[0] + a.j10 * acc[1]) + acc[6]);
        atomicAdd(p.dkd + o * 2 + 1, -(a.j01 * acc[0] + a.j11 * acc[1]) + acc[7]);
    }
    if (p.jd && (p.djd || p.djs)) {
        // J = S inv(D): dS = dJ inv(D)^T;  d inv = S^T dJ;  dD = -inv^T (d inv) inv^T
        const float a0 = p.jd[o * 4], a1 = p.jd[o * 4 + 1], a2 = p.jd[o * 4 + 2], a3 = p.jd[o * 4 + 3];
        const float idet = 1.f / (a0 * a3 - a1 * a2);
        const float i00 = a3 * idet, i01 = -a1 * idet, i10 = -a2 * idet, i11 = a0 * idet;
        const float s0 = p.js[o * 4], s1 = p.js[o * 4 + 1], s2 = p.js[o * 4 + 2], s3 = p.js[o * 4 + 3];
        const float d00 = acc[2], d01 = acc[3], d10 = acc[4], d11 = acc[5];                         // dJ
        if (p.djs) {
            atomicAdd(p.djs + o * 4, d00 * i00 + d01 * i01);
            atomicAdd(p.djs + o * 4 + 1, d00 * i10 + d01 * i11);
            atomicAdd(p.djs + o * 4 + 2, d10 * i00 + d11 * i01);
            atomicAdd(p.djs + o * 4 + 3, d10 * i10 + d11 * i11);
        }
        if (p.djd) {
            const float e00 = s0 * d00 + s2 * d10, e01 = s0 * d01 + s2 * d11, e10 = s1 * d00 + s3 * d10, e11 = s1 * d01 + s3 * d11;   // S^T dJ
            // t = inv^T e
            const float t00 = i00 * e00 + i10 * e10, t01 = i00 * e01 + i10 * e11, t10 = i01 * e00 + i11 * e10, t11 = i01 * e01 + i11 * e11;
            // dD = -t inv^T
            atomicAdd(p.djd + o * 4, -(t00 * i00 + t01 * i01));
            atomicAdd(p.djd + o * 4 + 1, -(t00 * i10 + t01 * i11));
            atomicAdd(p.djd + o * 4 + 2, -(t10 * i00 + t11 * i01));
            atomicAdd(p.djd + o * 4 + 3, -(t10 * i10 + t11 * i11));
        }
    }
}

// ------------------------------------------------------------------------------------------------ softmax_combine
constexpr int MAXK1 = 32;

__global__ __launch_bounds__(NT) void softmax_combine_fwd_kernel(const float* __restrict__ logit, int ldl, const float* __restrict__ motions,
                                                                 int ldm, int B, int HW, int K1, float* __restrict__ deform,
                                                                 float* __restrict__ mask, float* __restrict__ logit_nchw) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int b = (int)(i / HW), pix = (int)(i % HW);
        const float* l = logit + (size_t)i * ldl;
        float mxv = -INFINITY;
        for (int k = 0; k < K1; ++k) mxv = fmaxf(mxv, l[k]);
        float e[MAXK1], s = 0.f;
        for (int k = 0; k < K1; ++k) { e[k] = expf(l[k] - mxv); s += e[k]; }
        const float inv = 1.f / s;
        float dx = 0.f, dy = 0.f;
        for (int k = 0; k < K1; ++k) {
            const float mk = e[k] * inv;
            const float* mo = motions + ((size_t)(b * K1 + k) * HW + pix) * ldm;
            dx += mk * mo[0];
            dy += mk * mo[1];
            mask[((size_t)b * K1 + k) * HW + pix] = mk;
            logit_nchw[((size_t)b * K1 + k) * HW + pix] = l[k];
        }
        deform[(size_t)i * 2] = dx;
        deform[(size_t)i * 2 + 1] = dy;
    }
}

__global__ __launch_bounds__(NT) void softmax_combine_bwd_kernel(const float* __restrict__ motions, int ldm, int B, int HW, int K1,
                                                                 const float* __restrict__ mask, const float* __restrict__ ddeform,
                                                                 const float* __restrict__ dmask, const float* __restrict__ dlogit_nchw,
                                                                 float* __restrict__ dlogit, int lddl, float* __restrict__ dmotions) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int b = (int)(i / HW), pix = (int)(i % HW);
        const float gx = ddeform ? ddeform[(size_t)i * 2] : 0.f, gy = ddeform ? ddeform[(size_t)i * 2 + 1] : 0.f;
        float mk[MAXK1], dm[MAXK1], dot = 0.f;
        for (int k = 0; k < K1; ++k) {
            const size_t o = ((size_t)b * K1 + k) * HW + pix;
            const float* mo = motions + o * ldm;
            mk[k] = mask[o];
            dm[k] = gx * mo[0] + gy * mo[1] + (dmask ? dmask[o] : 0.f);
            dot += mk[k] * dm[k];
            if (dmotions) {
                dmotions[o * ldm] += mk[k] * gx;
                dmotions[o * ldm + 1] += mk[k] * gy;
            }
        }
        for (int k = 0; k < K1; ++k)
            dlogit[(size_t)i * lddl + k] += mk[k] * (dm[k] - dot) + (dlogit_nchw ? dlogit_nchw[((size_t)b * K1 + k) * HW + pix] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ kp_head
// one workgroup per (b, k): softmax over the H*W logits at temperature T, expectation of the [-1,1]^2 grid and of the 4 Jacobian maps
__global__ __launch_bounds__(NT) void kp_head_fwd_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ jm, int ldj,
                                                         int B, int H, int W, int K, float inv_temp, float* __restrict__ kp,
                                                         float* __restrict__ jac, float* __restrict__ stat) {
    __shared__ float lds[7 * 4];
    const int b = blockIdx.x / K, k = blockIdx.x % K, HW = H * W;
    const float* l = logits + (size_t)b * HW * ldl + k;
    float mxv = -INFINITY;
    for (int p = threadIdx.x; p < HW; p += NT) mxv = fmaxf(mxv, l[(size_t)p * ldl] * inv_temp);
    mxv = block_max(mxv, lds);
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < HW; p += NT) {
        const float e = expf(l[(size_t)p * ldl] * inv_temp - mxv);
        acc[0] += e;
        acc[1] += e * gcoord(p % W, W);
        acc[2] += e * gcoord(p / W, H);
        if (jm) {
            const float* j = jm + ((size_t)b * HW + p) * ldj;
            acc[3] += e * j[0]; acc[4] += e * j[1]; acc[5] += e * j[2]; acc[6] += e * j[3];
        }
    }
    block_reduce<7>(acc, lds);
    if (threadIdx.x == 0) {
        const float inv = 1.f / acc[0];
        const size_t o = (size_t)b * K + k;
        kp[o * 2] = acc[1] * inv;
        kp[o * 2 + 1] = acc[2] * inv;
        if (jm) { jac[o * 4] = acc[3] * inv; jac[o * 4 + 1] = acc[4] * inv; jac[o * 4 + 2] = acc[5] * inv; jac[o * 4 + 3] = acc[6] * inv; }
        stat[o * 2] = mxv;
        stat[o * 2 + 1] = inv;
    }
}

// one thread per (b, pixel): d logit[b,p,k] += p_k / T * (t_pk - E_k[t]) with t_pk = g_p . dkp_k + jm_p . djac_k and
// E_k[t] = kp_k . dkp_k + jac_k . djac_k (the forward outputs ARE the expectations);  d jm[b,p,:] += sum_k p_k djac_k
__global__ __launch_bounds__(NT) void kp_head_bwd_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ jm, int ldj, int B,
                                                         int H, int W, int K, float inv_temp, const float* __restrict__ kp,
                                                         const float* __restrict__ jac, const float* __restrict__ stat,
                                                         const float* __restrict__ dkp, const float* __restrict__ djac,
                                                         float* __restrict__ dlogits, int lddl, float* __restrict__ djm, int lddj) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        const float gx = gcoord(p % W, W), gy = gcoord(p / W, H);
        float j[4] = {0.f, 0.f, 0.f, 0.f}, dj[4] = {0.f, 0.f, 0.f, 0.f};
        if (jm) for (int q = 0; q < 4; ++q) j[q] = jm[(size_t)i * ldj + q];
        for (int k = 0; k < K; ++k) {
            const size_t o = (size_t)b * K + k;
            const float pk = expf(logits[(size_t)i * ldl + k] * inv_temp - stat[o * 2]) * stat[o * 2 + 1];
            const float gkx = dkp ? dkp[o * 2] : 0.f, gky = dkp ? dkp[o * 2 + 1] : 0.f;
            float t = gx * gkx + gy * gky - (kp[o * 2] * gkx + kp[o * 2 + 1] * gky);
            if (jm && djac) {
                for (int q = 0; q < 4; ++q) {
                    t += (j[q] - jac[o * 4 + q]) * djac[o * 4 + q];
                    dj[q] += pk * djac[o * 4 + q];
                }
            }
            dlogits[(size_t)i * lddl + k] += pk * inv_temp * t;
        }
        if (jm && djm && djac) for (int q = 0; q < 4; ++q) djm[(size_t)i * lddj + q] += dj[q];
    }
}

}  // namespace

extern "C" int mrfa_kp_gaussian_fwd(void* stream, const float* kp, const float* pos, int B, int K, int H, int W, float variance, float* out,
                                    int ldo) {
    MRFA_CHECK_ARG(kp && out && B > 0 && K > 0 && H > 1 && W > 1 && variance > 0.f, "kp_gaussian_fwd: bad arguments");
    const long long total = (long long)B * H * W * K;
    hipLaunchKernelGGL(kp_gaussian_fwd_kernel, dim3(stream_grid(total, NT)), dim3(NT), 0, (hipStream_t)stream, kp, pos, B, K, H, W,
                       1.f / variance, out, ldo);
    MRFA_CHECK_LAUNCH("mrfa_kp_gaussian_fwd");
    return 0;
}

extern "C" int mrfa_kp_gaussian_bwd(void* stream, const float* kp, int B, int K, int H, int W, float variance, const float* dout, int lddo,
                                    float* dkp, float* dpos) {
    MRFA_CHECK_ARG(kp && dout && B > 0 && K > 0 && H > 1 && W > 1 && variance > 0.f, "kp_gaussian_bwd: bad arguments");
    hipLaunchKernelGGL(kp_gaussian_bwd_kernel, dim3(B * K), dim3(NT), 0, (hipStream_t)stream, kp, B, K, H, W, 1.f / variance, dout, lddo, dkp,
                       dpos);
    MRFA_CHECK_LAUNCH("mrfa_kp_gaussian_bwd");
    return 0;
}

static int check_prior(const mrfa_prior_params& p, const char* what) {
    MRFA_CHECK_ARG(p.kd && p.ks && p.src && p.motions && p.B > 0 && p.K > 0 && p.H > 1 && p.W > 1 && p.C > 0 && p.C <= 8, "%s: bad arguments",
                   what);
    MRFA_CHECK_ARG((p.jd == nullptr) == (p.js == nullptr), "%s: jd and js come together", what);
    MRFA_CHECK_ARG(p.ldm >= 2 && p.lds >= p.C, "%s: bad leading dimensions", what);
    return 0;
}

extern "C" int mrfa_prior_motion_fwd(void* stream, const mrfa_prior_params* pp) {
    const mrfa_prior_params& p = *pp;
    if (int rc = check_prior(p, "prior_motion_fwd")) return rc;
    MRFA_CHECK_ARG(p.inp && p.ldi >= (p.K + 1) * (p.C + 1), "prior_motion_fwd: hourglass input buffer");
    const long long total = (long long)p.B * (p.K + 1) * p.H * p.W;
    hipLaunchKernelGGL(prior_motion_fwd_kernel, dim3(stream_grid(total, NT)), dim3(NT), 0, (hipStream_t)stream, p);
    MRFA_CHECK_LAUNCH("mrfa_prior_motion_fwd");
    return 0;
}

extern "C" int mrfa_prior_motion_bwd(void* stream, const mrfa_prior_params* pp) {
    const mrfa_prior_params& p = *pp;
    if (int rc = check_prior(p, "prior_motion_bwd")) return rc;
    MRFA_CHECK_ARG(p.dinp && p.lddi >= (p.K + 1) * (p.C + 1), "prior_motion_bwd: gradient of the hourglass input buffer");
    hipLaunchKernelGGL(prior_motion_bwd_kernel, dim3(p.B * (p.K + 1)), dim3(NT), 0, (hipStream_t)stream, p);
    MRFA_CHECK_LAUNCH("mrfa_prior_motion_bwd");
    return 0;
}

extern "C" int mrfa_softmax_combine_fwd(void* stream, const float* logit, int ldl, const float* motions, int ldm, int B, int H, int W, int K1,
                                        float* deformation, float* mask, float* logit_nchw) {
    MRFA_CHECK_ARG(logit && motions && deformation && mask && logit_nchw && K1 > 0 && K1 <= MAXK1, "softmax_combine_fwd: bad arguments (K1 <= %d)",
                   MAXK1);
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(softmax_combine_fwd_kernel, dim3(stream_grid(total, NT)), dim3(NT), 0, (hipStream_t)stream, logit, ldl, motions, ldm, B,
                       H * W, K1, deformation, mask, logit_nchw);
    MRFA_CHECK_LAUNCH("mrfa_softmax_combine_fwd");
    return 0;
}

extern "C" int mrfa_softmax_combine_bwd(void* stream, const float* motions, int ldm, int B, int H, int W, int K1, const float* mask,
                                        const float* ddeformation, const float* dmask, const float* dlogit_nchw, float* dlogit, int lddl,
                                        float* dmotions) {
    MRFA_CHECK_ARG(motions && mask && dlogit && K1 > 0 && K1 <= MAXK1, "softmax_combine_bwd: bad arguments");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(softmax_combine_bwd_kernel, dim3(stream_grid(total, NT)), dim3(NT), 0, (hipStream_t)stream, motions, ldm, B, H * W, K1,
                       mask, ddeformation, dmask, dlogit_nchw, dlogit, lddl, dmotions);
    MRFA_CHECK_LAUNCH("mrfa_softmax_combine_bwd");
    return 0;
}

extern "C" int mrfa_kp_head_fwd(void* stream, const float* logits, int ldl, const float* jm, int ldj, int B, int H, int W, int K,
                                float temperature, float* kp, float* jac, float* stat) {
    MRFA_CHECK_ARG(logits && kp && stat && B > 0 && K > 0 && H > 1 && W > 1 && temperature > 0.f, "kp_head_fwd: bad arguments");
    MRFA_CHECK_ARG(jm == nullptr || jac != nullptr, "kp_head_fwd: jacobian maps without an output");
    hipLaunchKernelGGL(kp_head_fwd_kernel, dim3(B * K), dim3(NT), 0, (hipStream_t)stream, logits, ldl, jm, ldj, B, H, W, K, 1.f / temperature, kp,
                       jac, stat);
    MRFA_CHECK_LAUNCH("mrfa_kp_head_fwd");
    return 0;
}

extern "C" int mrfa_kp_head_bwd(void* stream, const float* logits, int ldl, const float* jm, int ldj, int B, int H, int W, int K,
                                float temperature, const float* kp, const float* jac, const float* stat, const float* dkp, const float* djac,
                                float* dlogits, int lddl, float* djm, int lddj) {
    MRFA_CHECK_ARG(logits && kp && stat && dlogits && B > 0 && K > 0 && temperature > 0.f, "kp_head_bwd: bad arguments");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(kp_head_bwd_kernel, dim3(stream_grid(total, NT)), dim3(NT), 0, (hipStream_t)stream, logits, ldl, jm, ldj, B, H, W, K,
                       1.f / temperature, kp, jac, stat, dkp, djac, dlogits, lddl, djm, lddj);
    MRFA_CHECK_LAUNCH("mrfa_kp_head_bwd");
    return 0;
}
