// Micro-benchmark of the k3 tap-list kernel (csrc/uad_convk16.inc) outside the library: one launch shape, HIP-event timing, compile-time
// ablations (-DK3_ABL=<bits>: 1 no weight loads in the loop | 2 no LDS fragment reads | 4 no staging | 8 no MFMAs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DK3_ABL=n] tools/k3_ubench.hip -o tools/k3_ubench[_n]
//   tools/k3_ubench N H W CA Nn [planes]        (k3 s1 forward form)
#include <vector>
#include <string.h>
#include "../unsupervised_anomaly_detection_brain_mri_amd/csrc/uad_gemm_common.h"

namespace {
struct ConvWArgs {
    const float* big; const float* small_; float* partial; UadXform xfb, xfs; UadConvDesc d;
    int Mtot, Kt, kper, lws, lhs; unsigned long long* dbgbuf; int npl = 2; int abl = 0;
};
#include "../unsupervised_anomaly_detection_brain_mri_amd/csrc/uad_convk16.inc"
}

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 96, H = argc > 2 ? atoi(argv[2]) : 8, W = argc > 3 ? atoi(argv[3]) : 8;
    const int CA = argc > 4 ? atoi(argv[4]) : 512, Nn = argc > 5 ? atoi(argv[5]) : 512, planes = argc > 6 ? atoi(argv[6]) : 2;
    const char* mode = argc > 7 ? argv[7] : "f1";      // f1: k3 s1 forward form | f2: k3 s2 forward form (input 2H x 2W) | d2: k3 s2 transposed form (output 2H x 2W, four class launches)
    const bool f2 = !strcmp(mode, "f2"), d2 = !strcmp(mode, "d2"), k1 = !strcmp(mode, "k1");      // k1: k1 s1 convolution
    const size_t in_e = (size_t)N * H * W * CA * (f2 ? 4 : 1), out_e = (size_t)N * H * W * Nn * (d2 ? 4 : 1), w_e = (size_t)(k1 ? 1 : 9) * CA * Nn;
    float *in, *out; unsigned short* w16;
    CK_(hipMalloc(&in, in_e * 4)); CK_(hipMalloc(&out, out_e * 4)); CK_(hipMalloc(&w16, w_e * 2 * 3));
    std::vector<float> h(in_e);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    CK_(hipMemcpy(in, h.data(), in_e * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> hw(w_e * 3);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((s >> 12) & 0x1ff)); }       // bf16 values near 0.01 .. 0.03
    CK_(hipMemcpy(w16, hw.data(), w_e * 2 * 3, hipMemcpyHostToDevice));
    UadConvDesc d{N, H, W, CA, H, W, Nn, 3, 1, 1};
    if (f2) d = UadConvDesc{N, 2 * H, 2 * W, CA, H, W, Nn, 3, 2, 0};
    if (d2) d = UadConvDesc{N, 2 * H, 2 * W, Nn, H, W, CA, 3, 2, 0};
    if (k1) d = UadConvDesc{N, H, W, CA, H, W, Nn, 1, 1, 0};
    const bool ft = !d2;
    UadEpilogue ep; memset(&ep, 0, sizeof ep); ep.kind = UAD_EPI_BIAS;
    if (!strcmp(mode, "w1") || !strcmp(mode, "w2")) {
        // filter gradient of a k3 layer: big [N, S H, S W, CA] (x) small [N, H, W, Nn] -> slabs [splits][9 CA][Nn]
        const int S = mode[1] - '0';
        UadConvDesc dw{N, S * H, S * W, CA, H, W, Nn, 3, S, S == 1 ? 1 : 0};
        const WK3Choice k3 = choose_wk3(dw);
        if (!k3.ok) { fprintf(stderr, "shape not taken by the k3 filter-gradient kernel\n"); return 1; }
        const size_t big_e = (size_t)N * S * H * S * W * CA, small_e = (size_t)N * H * W * Nn, part_e = (size_t)k3.splits * 9 * CA * Nn;
        float *big, *sm, *part;
        CK_(hipMalloc(&big, big_e * 4)); CK_(hipMalloc(&sm, small_e * 4)); CK_(hipMalloc(&part, part_e * 4));
        std::vector<float> hb(big_e > small_e ? big_e : small_e);
        unsigned s2 = 777u;
        for (auto& v : hb) { s2 = s2 * 1664525u + 1013904223u; v = ((s2 >> 8) & 0xffff) / 65536.f - 0.5f; }
        CK_(hipMemcpy(big, hb.data(), big_e * 4, hipMemcpyHostToDevice));
        for (auto& v : hb) { s2 = s2 * 1664525u + 1013904223u; v = ((s2 >> 8) & 0xffff) / 65536.f - 0.5f; }
        CK_(hipMemcpy(sm, hb.data(), small_e * 4, hipMemcpyHostToDevice));
        ConvWArgs wa; memset(&wa, 0, sizeof wa);
        wa.big = big; wa.small_ = sm; wa.partial = part; wa.d = dw; wa.Mtot = 9 * CA; wa.Kt = N * H * W; wa.lws = wa.lhs = -1; wa.npl = 2;
        hipStream_t st; CK_(hipStreamCreate(&st));
        hipEvent_t a, b; CK_(hipEventCreate(&a)); CK_(hipEventCreate(&b));
        auto go = [&]() { if (S == 1) { if (k3.ncb == 2) launch_convk_w16<1, 2>(wa, k3, st); else launch_convk_w16<1, 1>(wa, k3, st); } else launch_convk_w16<2, 1>(wa, k3, st); };
        for (int i = 0; i < 3; ++i) go();
        CK_(hipStreamSynchronize(st));
        const int reps = 30;
        CK_(hipEventRecord(a, st));
        for (int i = 0; i < reps; ++i) go();
        CK_(hipEventRecord(b, st));
        CK_(hipEventSynchronize(b));
        float ms = 0; CK_(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1e3 / reps, flop = 2.0 * N * H * W * 9.0 * CA * Nn;
        printf("%s N=%d %dx%d CB=%d CS=%d ncb=%d splits=%d: %.1f us  %.1f TFLOP/s algorithmic  (x3 executed = %.3f of 2500)\n", mode, N, H, W, CA, Nn, k3.ncb, k3.splits, us,
               flop / us * 1e-6, flop * 3 / us * 1e-6 / 2500.0);
        std::vector<float> hp(part_e);
        CK_(hipMemcpy(hp.data(), part, part_e * 4, hipMemcpyDeviceToHost));
        unsigned long long x = 0; double sum = 0;
        for (size_t i = 0; i < part_e; ++i) { unsigned u; memcpy(&u, &hp[i], 4); x = x * 1099511628211ull ^ u; sum += hp[i]; }
        printf("   output digest %016llx  sum %.9g\n", x, sum);
        return 0;
    }
    hipStream_t st; CK_(hipStreamCreate(&st));
    hipEvent_t a, b; CK_(hipEventCreate(&a)); CK_(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) run_convk16(d, ft, in, out, ep, w16, (long long)w_e, planes, st);
    CK_(hipStreamSynchronize(st));
    const int reps = 50;
    CK_(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) run_convk16(d, ft, in, out, ep, w16, (long long)w_e, planes, st);
    CK_(hipEventRecord(b, st));
    CK_(hipEventSynchronize(b));
    float ms = 0; CK_(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, flop = 2.0 * N * H * W * (k1 ? 1.0 : 9.0) * CA * Nn;
    printf("K3_ABL=%d %s N=%d %dx%d CA=%d Nn=%d planes=%d: %.1f us  %.1f TFLOP/s algorithmic  (x%d executed = %.3f of 2500)\n", K3_ABL, mode, N, H, W, CA, Nn, planes, us,
           flop / us * 1e-6, planes == 3 ? 6 : 3, flop * (planes == 3 ? 6 : 3) / us * 1e-6 / 2500.0);
    std::vector<float> ho(out_e);
    CK_(hipMemcpy(ho.data(), out, out_e * 4, hipMemcpyDeviceToHost));
    unsigned long long x = 0; double sum = 0;
    for (size_t i = 0; i < out_e; ++i) { unsigned u; memcpy(&u, &ho[i], 4); x = x * 1099511628211ull ^ u; sum += ho[i]; }
    printf("   output digest %016llx  sum %.9g\n", x, sum);
    return 0;
}
