// conv_lean.hip against conv_small.hip / conv_halo.hip on the keypoint encoder's 3x3 shapes: per-launch time from a C++ launch loop over the ABI
// (HIP events, 200 back-to-back launches) and the largest difference between the kernels' outputs.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/ubench/lean_bench.cpp -Lmrfa_amd/_lib -lmrfa_hip -Wl,-rpath,'$ORIGIN/../../../mrfa_amd/_lib' -o tools/ubench/bin/lean_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mrfa_hip.h"

static float* dalloc(size_t n, float v, unsigned seed) {
    float* p;
    hipMalloc(&p, n * 4);
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = v * ((float)(s >> 8) / 16777216.f - 0.5f); }
    hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}

// GEMM=1: gemm_lean_kernel against whatever the dispatcher picks without it (conv_small / row-tiled) on the encoder's 1x1 shapes
static int gemm_main() {
    struct G { const char* name; long long M; int W, Cin, Cout; };
    const G gs[] = {{"tokens 192->576", 4416, 276, 192, 576}, {"tokens 576->192", 4416, 276, 576, 192}, {"tokens 192->192", 4416, 276, 192, 192},
                    {"layer1 64->256 @64^2", 65536, 64, 64, 256}, {"layer1 256->64 @64^2", 65536, 64, 256, 64}, {"layer1 64->64 @64^2", 65536, 64, 64, 64},
                    {"fuse 64->32 @32^2", 16384, 32, 64, 32}, {"fuse 128->32 @16^2", 4096, 16, 128, 32}, {"fuse 128->64 @16^2", 4096, 16, 128, 64}};
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 200;
    mrfa_set_mfma_mode(1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s %-9s | %9s %7s | %9s %7s %10s | max |lean - other| (scale)\n", "shape", "variant", "lean us", "TF/s", "other us", "TF/s", "config");
    for (const G& g : gs) {
        const int Ci = g.Cin, Co = g.Cout, cop = (Co + 127) / 128 * 128;
        const int N = (int)(g.M / g.W / (g.W == 276 ? 1 : g.W)), H = g.W == 276 ? 1 : g.W;
        float* x = dalloc(g.M * Ci, 2.f, 1);
        float* wo = dalloc((size_t)Co * Ci, 0.2f, 2);
        float *w, *y[2];
        hipMalloc(&w, (size_t)cop * Ci * 4); hipMemset(w, 0, (size_t)cop * Ci * 4);
        mrfa_pack_conv_weight(nullptr, wo, w, Co, Ci, 1, 1, 0);
        short* wsb; const long long piece = (long long)cop * Ci;
        hipMalloc(&wsb, 3 * piece * 2); hipMemset(wsb, 0, 3 * piece * 2);
        mrfa_pack_desc d; memset(&d, 0, sizeof(d));
        d.src = wo; d.dst[0] = (float*)wsb; d.mode[0] = 8; d.ndst = 1; d.Cout = Co; d.Cin = Ci; d.R = 1; d.S = 1;
        if (mrfa_pack_conv_weights_multi(nullptr, &d, 1)) { printf("pack: %s\n", mrfa_last_error()); return 1; }
        for (int k = 0; k < 2; ++k) { hipMalloc(&y[k], g.M * Co * 4); hipMemset(y[k], 0, g.M * Co * 4); }
        float* res = dalloc(g.M * Co, 1.f, 3); float* bias = dalloc(Co, 1.f, 6);
        double* stats; const size_t sbytes = (size_t)3 * MRFA_STATS_SLOTS * 2 * Co * 8;
        hipMalloc(&stats, sbytes); hipMemset(stats, 0, sbytes);
        unsigned* tickets; hipMalloc(&tickets, (size_t)(iters + 8) * MRFA_FIN_WORDS * 4);
        float* fin[6]; for (int k = 0; k < 6; ++k) fin[k] = dalloc(3 * Co, 1.f, 10 + k);
        for (int variant = 0; variant < 2; ++variant) {        // 0: bias + residual (the token linears); 1: statistics + finalize (the 1x1 conv + BN layers)
            mrfa_conv_params p; memset(&p, 0, sizeof(p));
            p.x = x; p.ldx = Ci; p.Hin = H; p.Win = g.W; p.N = N; p.Cin = Ci;
            p.w = w; p.w_ld = Ci; p.w_tap = (long long)cop * Ci; p.w_rows = cop; p.w_split = wsb; p.w_piece = piece;
            p.ldy = Co; p.Cout = Co; p.Hout = H; p.Wout = g.W; p.R = 1; p.S = 1; p.pad = 0; p.alpha = 1.f; p.nbatch = 1;
            if (variant == 0) { p.bias = bias; p.res = res; p.ldr = Co; }
            else {
                p.stats = stats; p.groups = 1;
                p.fin_gamma = fin[0]; p.fin_beta = fin[1]; p.fin_momentum = 0.1f; p.fin_eps = 1e-5f; p.fin_count = g.M;
                p.fin_scale = fin[2]; p.fin_shift = fin[3]; p.fin_mean = fin[4]; p.fin_invstd = fin[5]; p.fin_counter = tickets;
            }
            float us[2] = {0, 0}; int cfg[2] = {0, 0};
            for (int k = 0; k < 2; ++k) {
                mrfa_set_tuning("gemm_lean", k == 0);
                p.y = y[k];
                hipMemset(tickets, 0, (size_t)(iters + 8) * MRFA_FIN_WORDS * 4);
                for (int i = 0; i < iters + 5; ++i) {
                    if (i == 5) { hipDeviceSynchronize(); hipEventRecord(e0); }
                    if (p.fin_scale) p.fin_counter = tickets + (size_t)i * MRFA_FIN_WORDS;
                    if (mrfa_conv2d_nhwc(nullptr, &p)) { printf("error: %s\n", mrfa_last_error()); return 1; }
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                us[k] = 1e3f * ms / iters; cfg[k] = mrfa_conv2d_last_config();
                if (k == 0 && !(cfg[0] & (1 << 26))) us[0] = -1.f;
            }
            mrfa_set_tuning("gemm_lean", 1);
            double md = 0, sc = 0;
            std::vector<float> a(g.M * Co), b(g.M * Co);
            hipMemcpy(a.data(), y[0], g.M * Co * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y[1], g.M * Co * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); sc = fmax(sc, fabs((double)b[i])); }
            const double fl = 2.0 * g.M * Ci * (double)Co;
            printf("%-22s %-9s | %9.2f %7.1f | %9.2f %7.1f 0x%08x | %.3e (%.2f)\n", g.name, variant ? "stats+fin" : "bias+res", us[0], us[0] > 0 ? fl / us[0] / 1e6 : 0, us[1], fl / us[1] / 1e6,
                   cfg[1], md, sc);
        }
        hipFree(x); hipFree(wo); hipFree(w); hipFree(wsb); hipFree(y[0]); hipFree(y[1]); hipFree(res); hipFree(bias); hipFree(stats); hipFree(tickets);
        for (int k = 0; k < 6; ++k) hipFree(fin[k]);
    }
    return 0;
}

int main() {
    if (getenv("GEMM")) return gemm_main();
    struct Shape { const char* name; int H, W, C; };
    const Shape shapes[] = {{"32->32 @64^2", 64, 64, 32}, {"64->64 @32^2", 32, 32, 64}, {"128->128 @16^2", 16, 16, 128}};
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 200;
    mrfa_set_mfma_mode(1);
    if (getenv("GEO")) mrfa_set_tuning("conv_lean_geo", atoi(getenv("GEO")));      // one geometry of conv_lean.hip's table (shapes it does not take print -1)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int only_lean = getenv("ONLY_LEAN") ? 1 : 0;     // PMC passes: one kernel per run
    printf("%-16s %3s %-9s | %9s %7s | %9s %7s | %9s %7s | max |lean - small| (scale)\n", "shape", "N", "variant", "lean us", "TF/s", "small us", "TF/s", "halo/tile", "TF/s");
    for (const Shape& s : shapes) {
        for (int N : {8, 16, 24}) {
            if (getenv("NONLY") && atoi(getenv("NONLY")) != N) continue;
            const long long M = (long long)N * s.H * s.W;
            const int Cc = s.C, cop = 128;
            float* x = dalloc(M * Cc, 2.f, 1);
            float* wo = dalloc((size_t)Cc * Cc * 9, 0.2f, 2);        // OIHW
            float *w, *y[3], *res, *bx, *vecs;
            hipMalloc(&w, (size_t)9 * cop * Cc * 4); hipMemset(w, 0, (size_t)9 * cop * Cc * 4);
            mrfa_pack_conv_weight(nullptr, wo, w, Cc, Cc, 3, 3, 0);
            short* wsb; const long long piece = (long long)9 * cop * Cc;
            hipMalloc(&wsb, 3 * piece * 2); hipMemset(wsb, 0, 3 * piece * 2);
            mrfa_pack_desc d; memset(&d, 0, sizeof(d));
            d.src = wo; d.dst[0] = (float*)wsb; d.mode[0] = 8; d.ndst = 1; d.Cout = Cc; d.Cin = Cc; d.R = 3; d.S = 3;
            if (mrfa_pack_conv_weights_multi(nullptr, &d, 1)) { printf("pack: %s\n", mrfa_last_error()); return 1; }
            for (int k = 0; k < 3; ++k) { hipMalloc(&y[k], M * Cc * 4); hipMemset(y[k], 0, M * Cc * 4); }
            res = dalloc(M * Cc, 1.f, 3); bx = dalloc(M * Cc, 1.f, 4); vecs = dalloc(16 * Cc * 3, 1.f, 5);
            double* stats; const size_t sbytes = (size_t)3 * MRFA_STATS_SLOTS * 2 * Cc * 8;
            hipMalloc(&stats, sbytes); hipMemset(stats, 0, sbytes);
            unsigned* tickets; hipMalloc(&tickets, (size_t)(iters + 8) * MRFA_FIN_WORDS * 4);        // one finalize ticket block per launch, zeroed outside the timed loop
            float *fin[6]; for (int k = 0; k < 6; ++k) fin[k] = dalloc(3 * Cc, 1.f, 10 + k);
            for (int variant = 0; variant < 3; ++variant) {       // 0: forward conv + statistics + finalize; 1: + prologue + residual + ReLU; 2: data gradient with bst_*
                mrfa_conv_params p; memset(&p, 0, sizeof(p));
                p.x = x; p.ldx = Cc; p.Hin = s.H; p.Win = s.W; p.N = N; p.Cin = Cc;
                p.w = w; p.w_ld = Cc; p.w_tap = (long long)cop * Cc; p.w_rows = cop; p.w_split = wsb; p.w_piece = piece;
                p.ldy = Cc; p.Cout = Cc; p.Hout = s.H; p.Wout = s.W; p.R = 3; p.S = 3; p.pad = 1; p.alpha = 1.f; p.nbatch = 1;
                p.stats = stats; p.groups = N >= 16 ? N / 8 : 1;
                if (variant <= 1) {
                    p.fin_gamma = fin[0]; p.fin_beta = fin[1]; p.fin_momentum = 0.1f; p.fin_eps = 1e-5f; p.fin_count = M / (p.groups > 1 ? p.groups : 1);
                    p.fin_scale = fin[2]; p.fin_shift = fin[3]; p.fin_mean = fin[4]; p.fin_invstd = fin[5];
                    p.fin_counter = tickets;
                }
                if (variant == 1) { p.in_scale = vecs; p.in_shift = vecs + Cc; p.in_relu = 1; p.res = res; p.ldr = Cc; p.relu = 1; }
                if (variant == 2) { p.bst_x = bx; p.bst_ldx = Cc; p.bst_scale = vecs; p.bst_shift = vecs + 4 * Cc; p.bst_mean = vecs + 8 * Cc; p.bst_invstd = vecs + 12 * Cc; p.bst_relu = 1; }
                float us[3] = {0, 0, 0};
                bool small_ok = true;
                for (int k = 0; k < 3; ++k) {                      // 0: lean; 1: conv_small; 2: whatever is left (patch-tiled / row-tiled)
                    if (only_lean && k) continue;
                    if (variant == 1 && k == 1) continue;           // (conv_small has no prologue)
                    if (variant == 2 && k == 2) continue;           // (bst_* exists in the small-problem kernels only)
                    mrfa_set_tuning("conv_lean", k == 0);
                    mrfa_set_tuning("conv_small", k <= 1);
                    p.y = y[k];
                    if (variant == 2 && !mrfa_conv2d_bwdstats_supported(&p)) { if (k == 1) small_ok = false; continue; }   // (past conv_small's limits)
                    hipMemset(tickets, 0, (size_t)(iters + 8) * MRFA_FIN_WORDS * 4);
                    for (int i = 0; i < iters + 5; ++i) {
                        if (i == 5) { hipDeviceSynchronize(); hipEventRecord(e0); }
                        if (p.fin_scale) p.fin_counter = tickets + (size_t)i * MRFA_FIN_WORDS;
                        if (mrfa_conv2d_nhwc(nullptr, &p)) { printf("error: %s\n", mrfa_last_error()); return 1; }
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    us[k] = 1e3f * ms / iters;
                    if (k == 0 && !(mrfa_conv2d_last_config() & (1 << 27))) us[k] = -1.f;
                }
                mrfa_set_tuning("conv_lean", 1); mrfa_set_tuning("conv_small", 1);
                double md = 0, sc = 0;
                if (!only_lean && small_ok) {
                    const int other = variant == 1 ? 2 : 1;
                    std::vector<float> a(M * Cc), b(M * Cc);
                    hipMemcpy(a.data(), y[0], M * Cc * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y[other], M * Cc * 4, hipMemcpyDeviceToHost);
                    for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); sc = fmax(sc, fabs((double)b[i])); }
                }
                const double fl = 2.0 * M * Cc * (double)Cc * 9;
                const char* vn[3] = {"fwd+fin", "pro+res", "dgrad+bst"};
                printf("%-16s %3d %-9s | %9.2f %7.1f | %9.2f %7.1f | %9.2f %7.1f | %.3e (%.2f)\n", s.name, N, vn[variant], us[0], us[0] > 0 ? fl / us[0] / 1e6 : 0, us[1],
                       us[1] > 0 ? fl / us[1] / 1e6 : 0, us[2], us[2] > 0 ? fl / us[2] / 1e6 : 0, md, sc);
            }
            hipFree(x); hipFree(wo); hipFree(w); hipFree(wsb); for (int k = 0; k < 3; ++k) hipFree(y[k]);
            hipFree(res); hipFree(bx); hipFree(vecs); hipFree(stats); hipFree(tickets); for (int k = 0; k < 6; ++k) hipFree(fin[k]);
        }
    }
    return 0;
}
