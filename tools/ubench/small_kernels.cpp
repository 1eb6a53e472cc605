// True per-launch time of mrfa_conv2d_nhwc / mrfa_conv2d_wgrad_nhwc on the MTIA prior's small shapes: 200 back-to-back launches from C++
// (no Python launch overhead), HIP events.  Links libmrfa_hip.so.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/ubench/small_kernels.cpp -Lmrfa_amd/_lib -lmrfa_hip -Wl,-rpath,'$ORIGIN/../../../mrfa_amd/_lib' -o tools/ubench/bin/small_kernels
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mrfa_hip.h"

static float* dalloc(size_t n, float v) {
    float* p;
    hipMalloc(&p, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u) % 1000) / 1000.f - v * 0.5f;
    hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}

int main() {
    struct Shape { const char* name; int N, H, W, Cin, Cout, R; };
    const Shape shapes[] = {{"hr 32->32 3x3 @64", 8, 64, 64, 32, 32, 3}, {"hr 64->64 3x3 @32", 8, 32, 32, 64, 64, 3},
                            {"hr 128->128 3x3 @16", 8, 16, 16, 128, 128, 3}, {"hr 64->64 3x3 @64 (layer1)", 8, 64, 64, 64, 64, 3}, {"vit 192->576 1x1 tokens", 8, 1, 276, 192, 576, 1},
                            {"vit 576->192 1x1 tokens", 8, 1, 276, 576, 192, 1}, {"vit 192->192 1x1 tokens", 8, 1, 276, 192, 192, 1},
                            {"hr 64->256 1x1 @64", 8, 64, 64, 64, 256, 1},
                            {"gen 128->64 3x3 @256", 8, 256, 256, 128, 64, 3}, {"gen 256->128 3x3 @128", 8, 128, 128, 256, 128, 3},
                            {"gen 512->512 3x3 @32", 8, 32, 32, 512, 512, 3}, {"hg 256->512 3x3 @8", 8, 8, 8, 256, 512, 3}};
    mrfa_set_mfma_mode(getenv("MFMA") ? atoi(getenv("MFMA")) : 1);      // the product's default: bf16x6 on the 128-row tiles
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 200;
    printf("%-28s %10s %8s | %10s %8s\n", "shape", "conv us", "TF/s", "wgrad us", "TF/s");
    const int nmul = getenv("NMUL") ? atoi(getenv("NMUL")) : 1;          // batch multiplier (how the kernels scale with rows)
    const int only = getenv("SHAPE") ? atoi(getenv("SHAPE")) : -1;       // run one row of the table (PMC passes)
    int shape_idx = -1;
    for (Shape s : shapes) {
        if (++shape_idx != only && only >= 0) continue;
        s.N *= nmul;
        const long long M = (long long)s.N * s.H * s.W;
        const int T = s.R * s.R, cop = (s.Cout + 127) / 128 * 128;
        float* x = dalloc(M * s.Cin, 1.f);
        float* w = dalloc((size_t)T * cop * s.Cin, 0.2f);
        float* y = dalloc(M * s.Cout, 1.f);
        float* dw = dalloc((size_t)T * s.Cout * s.Cin, 0.f);
        double* stats; hipMalloc(&stats, MRFA_STATS_SLOTS * 2 * s.Cout * 8); hipMemset(stats, 0, MRFA_STATS_SLOTS * 2 * s.Cout * 8);
        mrfa_conv_params p;
        memset(&p, 0, sizeof(p));
        p.x = x; p.ldx = s.Cin; p.Hin = s.H; p.Win = s.W; p.N = s.N; p.Cin = s.Cin;
        p.w = w; p.w_ld = s.Cin; p.w_tap = (long long)cop * s.Cin; p.w_rows = cop;
        p.y = y; p.ldy = s.Cout; p.Cout = s.Cout; p.Hout = s.H; p.Wout = s.W; p.R = s.R; p.S = s.R; p.pad = s.R / 2;
        p.alpha = 1.f; p.nbatch = 1; p.stats = getenv("NOSTATS") ? nullptr : stats;
        mrfa_wgrad_params q;
        memset(&q, 0, sizeof(q));
        q.x = x; q.ldx = s.Cin; q.Hin = s.H; q.Win = s.W; q.N = s.N; q.Cin = s.Cin;
        q.dy = y; q.ldy = s.Cout; q.Cout = s.Cout; q.Hout = s.H; q.Wout = s.W; q.R = s.R; q.S = s.R; q.pad = s.R / 2;
        q.dw = dw; q.alpha = 1.f; q.nbatch = 1;
        float ms[2];
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 5; ++i) { if (which == 0) mrfa_conv2d_nhwc(nullptr, &p); else mrfa_conv2d_wgrad_nhwc(nullptr, &q); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) {
                int rc = which == 0 ? mrfa_conv2d_nhwc(nullptr, &p) : mrfa_conv2d_wgrad_nhwc(nullptr, &q);
                if (rc) { printf("error: %s\n", mrfa_last_error()); return 1; }
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[which], e0, e1);
        }
        const double fl = 2.0 * M * s.Cout * (double)s.Cin * T;
        // throughput form: 28 copies of the problem (own dW each) per mrfa_conv2d_wgrad_multi call -- what one problem costs once launch overhead is amortised
        float ms_multi = 0.f;
        {
            const int NC = 28;
            std::vector<mrfa_wgrad_params> qs(NC, q);
            float* dws; hipMalloc(&dws, (size_t)NC * T * s.Cout * s.Cin * 4); hipMemset(dws, 0, (size_t)NC * T * s.Cout * s.Cin * 4);
            for (int i = 0; i < NC; ++i) qs[i].dw = dws + (size_t)i * T * s.Cout * s.Cin;
            for (int i = 0; i < 3; ++i) mrfa_conv2d_wgrad_multi(nullptr, qs.data(), NC);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int it2 = iters / 10 > 0 ? iters / 10 : 1;
            for (int i = 0; i < it2; ++i) mrfa_conv2d_wgrad_multi(nullptr, qs.data(), NC);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_multi, e0, e1);
            ms_multi /= (float)(it2 * NC);
            hipFree(dws);
        }
        printf("%-28s %10.2f %8.1f | %10.2f %8.1f | multi %6.2f us/problem %6.1f TF/s\n", s.name, 1e3 * ms[0] / iters, fl / (ms[0] / iters) / 1e9, 1e3 * ms[1] / iters,
               fl / (ms[1] / iters) / 1e9, 1e3 * ms_multi, fl / ms_multi / 1e9);
        if (getenv("BN") && s.Cin == s.Cout) {                  // BatchNorm apply / two-phase backward on the same activation
            float *sc = dalloc(s.Cout, 1.f), *sh = dalloc(s.Cout, 1.f), *dx = dalloc(M * s.Cout, 0.f), *dg = dalloc(s.Cout, 0.f), *db = dalloc(s.Cout, 0.f);
            double* red; hipMalloc(&red, MRFA_STATS_SLOTS * 2 * s.Cout * 8); hipMemset(red, 0, MRFA_STATS_SLOTS * 2 * s.Cout * 8);
            mrfa_bnact_params a; memset(&a, 0, sizeof(a));
            a.x = x; a.ldx = s.Cin; a.N = s.N; a.H = s.H; a.W = s.W; a.C = s.Cin; a.scale = sc; a.shift = sh; a.relu = 1; a.y = y; a.ldy = s.Cout;
            mrfa_bnbwd_params b; memset(&b, 0, sizeof(b));
            b.x = x; b.ldx = s.Cin; b.N = s.N; b.H = s.H; b.W = s.W; b.C = s.Cin; b.scale = sc; b.shift = sh; b.relu = 1; b.mean = sh; b.invstd = sc;
            b.gamma = sc; b.dy = y; b.lddy = s.Cout; b.red = red; b.dx = dx; b.lddx = s.Cin; b.dgamma = dg; b.dbeta = db; b.train = 1; b.dx_overwrite = 1;
            float t[3];
            for (int which = 0; which < 3; ++which) {
                b.phase = which == 1 ? 1 : 2;
                for (int i = 0; i < iters + 5; ++i) {
                    if (i == 5) { hipDeviceSynchronize(); hipEventRecord(e0); }
                    if (which == 0) mrfa_bn_act_fwd(nullptr, &a); else mrfa_bn_act_bwd(nullptr, &b);
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t[which], e0, e1);
            }
            printf("    bn_act fwd %.2f us, bwd phase 1 %.2f us, phase 2 %.2f us\n", 1e3 * t[0] / iters, 1e3 * t[1] / iters, 1e3 * t[2] / iters);
            hipFree(sc); hipFree(sh); hipFree(dx); hipFree(dg); hipFree(db); hipFree(red);
        }
        hipFree(x); hipFree(w); hipFree(y); hipFree(dw); hipFree(stats);
    }
    return 0;
}
