// Implicit-GEMM convolution kernels for gfx950 (CDNA4), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s chip peak).
//
// Three kernel families cover every contraction on the AE/VAE hot path
// (reference: models/customlayers.py:16-38 Conv2D / Conv2DTranspose k5 s2 SAME, the 1x1 convs and
//  Dense layers of models/variational_autoencoder.py:20-36, and their tf.gradients):
//   F ("gather")  : small[n,i,j,cs]            = sum_tap,cb big[n,S*i-P+ky,S*j-P+kx,cb] * W[tap][cb][cs]
//   D ("scatter") : big[n,S*i-P+ky,S*j-P+kx,cb] += small[n,i,j,cs] * W[tap][cb][cs]   (per output-parity class)
//   W ("filter")  : dW[tap][cb][cs]             = sum_n,i,j big[..tap..,cb] * small[n,i,j,cs]
// Layout: NHWC activations, W[tap][cb][cs] weights (HWIO for Conv2D, [kh,kw,Cout,Cin] for Conv2DTranspose),
// so the channel (contraction) axis is always contiguous: 16-byte coalesced global loads, one 128-B line per
// pixel per 32 channels.  Tiles are staged global -> VGPR -> LDS (double buffered, one barrier per K-step) because
// the producer layer's frozen-BN affine + LeakyReLU is applied on load (activation tensors are stored once, pre-BN).
// The 64-lane wavefront owns (32*FM)x(32*FN) of the block tile; K is walked 8 at a time with the lane-half
// permutation k = 8*kk + 4*(lane>>5) + s so that one ds_read_b128 feeds four MFMAs.
#include <type_traits>
#include "uad_kernels.h"
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <array>
#include <string>
#include <vector>

#include "uad_gemm_common.h"

namespace {

constexpr int XF_LDS_CH = 512;  // activation-on-load tables (gamma', beta) staged once per workgroup

template <int BM, int BN, int BK, int WGM, int WGN, int KIND>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_gemm_kernel(const ConvGemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
    static_assert(BK % 8 == 0, "BK multiple of 8");
    constexpr int NKK = BK / 8;
    constexpr int LDA = BK + 4;  // (LDA/4) odd -> conflict-free ds_read_b128 across 16-lane groups
    constexpr int LDB = (KIND == KIND_F) ? (BN + 4) : (BK + 4);
    constexpr int A_EL = BM * LDA;
    constexpr int B_EL = (KIND == KIND_F) ? (BK * LDB) : (BN * LDB);
    constexpr int STAGE = A_EL + B_EL;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    __shared__ __attribute__((aligned(16))) float s_xf[2 * XF_LDS_CH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const UadConvDesc& d = a.d;
    const int S = d.S, P = d.P, KS = d.KS;

    // ---- tap set: all KS*KS taps (F) or the taps of this output-parity class (D) ----
    int py = 0, px = 0, ky0 = 0, kx0 = 0, nty = KS, ntx = KS, dy0 = 0, dx0 = 0;
    const int nsplit = a.nsplit;
    const int split = (KIND == KIND_D) ? (int)(blockIdx.z % nsplit) : (int)blockIdx.z;
    if (KIND == KIND_D) {
        const int cz = blockIdx.z / nsplit;
        py = (S - 1) - cz / S;
        px = (S - 1) - cz % S;
        ky0 = (py + P) % S;
        kx0 = (px + P) % S;
        dy0 = (py + P) / S;
        dx0 = (px + P) / S;
        nty = ky0 < KS ? (KS - ky0 + S - 1) / S : 0;
        ntx = kx0 < KS ? (KS - kx0 + S - 1) / S : 0;
    }
    const int ntaps = nty * ntx;
    const int ncc = a.CA / BK;
    const int nk = ntaps * ncc;
    // split-K: this workgroup contracts K-steps [ks0, ks1)
    const int kper = (nk + nsplit - 1) / nsplit;
    const int ks0 = split * kper;
    const int ks1 = min(nk, ks0 + kper);

    // ---- activation-on-load tables -> LDS (once) ----
    const bool xf = a.xf.scale != nullptr;
    if (xf) {
        for (int c = tid; c < a.CA; c += NT) {
            s_xf[c] = a.xf.scale[c] * a.xf.mult;
            s_xf[XF_LDS_CH + c] = a.xf.shift[c];
        }
    }

    // ---- per-thread A rows (positions) ----
    constexpr int TPR = BK / 4, RPP = NT / TPR, PA = (BM + RPP - 1) / RPP;
    const int acol = (tid % TPR) * 4;
    const int arow0 = tid / TPR;
    int rbase[PA], ry[PA], rx[PA];
    bool rok[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int r = arow0 + q * RPP;
        const int m = m0 + r;
        rok[q] = (r < BM) && (m < a.M);
        int n, i, j;
        decode_pos(rok[q] ? m : 0, d.HS, d.WS, a.lhs, a.lws, n, i, j);
        if (KIND == KIND_F) {
            rbase[q] = n * d.HB * d.WB;
            ry[q] = S * i - P;
            rx[q] = S * j - P;
        } else {
            rbase[q] = n * d.HS * d.WS;
            ry[q] = i;
            rx[q] = j;
        }
    }
    const int AH = (KIND == KIND_F) ? d.HB : d.HS;
    const int AW = (KIND == KIND_F) ? d.WB : d.WS;

    // ---- per-thread B slots (predicates are loop invariant; out-of-range slots read element 0 and are zeroed) ----
    constexpr int TPRB = (KIND == KIND_F) ? (BN / 4) : (BK / 4);
    constexpr int RPB = NT / TPRB;
    constexpr int BROWS = (KIND == KIND_F) ? BK : BN;
    constexpr int PB = (BROWS + RPB - 1) / RPB;
    const int bcol = (tid % TPRB) * 4;
    const int brow0 = tid / TPRB;
    bool bok[PB];
    int boff[PB];   // loop-invariant part of the weight offset
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int r = brow0 + q * RPB;
        if (KIND == KIND_F) {
            bok[q] = (r < BK) && (n0 + bcol) < a.Nn;
            boff[q] = bok[q] ? (r * d.CS + n0 + bcol) : 0;
        } else {
            bok[q] = (r < BN) && (n0 + r) < a.Nn;
            boff[q] = bok[q] ? ((n0 + r) * d.CS + bcol) : 0;
        }
    }

    float4 va[PA], vb[PB];
    unsigned okbits = 0;
    int xc0 = 0;

    // branch-free tile fetch: every load is issued unconditionally from a clamped (always valid) address
    auto load_tiles = [&](int tap, int c0) {
        int ky, kx, oy, ox;
        {
            const int ty = tap / ntx, tx = tap - ty * ntx;
            if (KIND == KIND_F) { ky = ty; kx = tx; oy = ky; ox = kx; }
            else { ky = ky0 + S * ty; kx = kx0 + S * tx; oy = dy0 - ty; ox = dx0 - tx; }
        }
        xc0 = c0;
        okbits = 0;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int y = ry[q] + oy, x = rx[q] + ox;
            const bool ok = rok[q] && (unsigned)y < (unsigned)AH && (unsigned)x < (unsigned)AW;
            const int pix = ok ? (rbase[q] + y * AW + x) : 0;
            va[q] = *reinterpret_cast<const float4*>(a.A + (size_t)pix * a.CA + c0 + acol);
            okbits |= (ok ? 1u : 0u) << q;
        }
        const int tapw = ky * KS + kx;
        const size_t wbase = (KIND == KIND_F) ? ((size_t)tapw * d.CB + c0) * d.CS : ((size_t)tapw * d.CB) * d.CS + c0;
#pragma unroll
        for (int q = 0; q < PB; ++q) vb[q] = *reinterpret_cast<const float4*>(a.W + wbase + boff[q]);
    };
    auto store_tiles = [&](int buf) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + A_EL;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xf) {
            sc = *reinterpret_cast<const float4*>(s_xf + xc0 + acol);
            sh = *reinterpret_cast<const float4*>(s_xf + XF_LDS_CH + xc0 + acol);
        }
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = arow0 + q * RPP;
            float4 v = va[q];
            if (xf) v = xform4(v, sc, sh, a.xf.alpha);
            // padding pixels stay exactly 0 (the pad is applied to the ACTIVATED tensor)
            v = keep4((okbits >> q) & 1u, v);
            if (r < BM) *reinterpret_cast<float4*>(sA + r * LDA + acol) = v;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int r = brow0 + q * RPB;
            const float4 v = keep4(bok[q], vb[q]);
            if (r < BROWS) *reinterpret_cast<float4*>(sB + r * LDB + bcol) = v;
        }
    };

    v16f acc[FM][FN];
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[im][jn][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;

    __syncthreads();   // s_xf visible
    int tap = ks0 / ncc, cc = ks0 - tap * ncc;
    if (ks0 < ks1) {
        load_tiles(tap, cc * BK);
        store_tiles(0);
    }
    __syncthreads();

    // fragment double buffer: the ds_reads of k-slice kk+1 are in flight while the MFMAs of kk issue
    float4 af[2][FM];
    float bf[2][FN][4];
    auto load_frags = [&](int slot, const float* sA, const float* sB, int kk) {
#pragma unroll
        for (int im = 0; im < FM; ++im)
            af[slot][im] = *reinterpret_cast<const float4*>(sA + (wm * WTM + im * 32 + l31) * LDA + kk * 8 + 4 * lh);
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            if (KIND == KIND_F) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    bf[slot][jn][s] = sB[(kk * 8 + 4 * lh + s) * LDB + wn * WTN + jn * 32 + l31];
            } else {
                const float4 t = *reinterpret_cast<const float4*>(sB + (wn * WTN + jn * 32 + l31) * LDB + kk * 8 + 4 * lh);
                bf[slot][jn][0] = t.x; bf[slot][jn][1] = t.y; bf[slot][jn][2] = t.z; bf[slot][jn][3] = t.w;
            }
        }
    };

    for (int ks = ks0; ks < ks1; ++ks) {
        const int buf = (ks - ks0) & 1;
        int ntap = tap, ncc_ = cc + 1;
        if (ncc_ == ncc) { ncc_ = 0; ntap = tap + 1; }
        const bool more = (ks + 1 < ks1);
        const float* sA = smem + buf * STAGE;
        const float* sB = sA + A_EL;
        load_frags(0, sA, sB, 0);
        if (more) load_tiles(ntap, ncc_ * BK);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < NKK) load_frags(cur ^ 1, sA, sB, kk + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int im = 0; im < FM; ++im) {
                    const float av = (s == 0) ? af[cur][im].x : (s == 1) ? af[cur][im].y : (s == 2) ? af[cur][im].z : af[cur][im].w;
#pragma unroll
                    for (int jn = 0; jn < FN; ++jn)
                        acc[im][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[cur][jn][s], acc[im][jn], 0, 0, 0);
                }
            }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
        tap = ntap;
        cc = ncc_;
    }

    // ---------------------------------------------------------------- epilogue
    if (nsplit > 1) {
        // raw partial tile into this split's slab (output layout); bias / activation-backward / column sums are
        // applied by splitk_epilogue_kernel after the slabs are summed in a fixed order
        float* slab = a.Out + (size_t)split * a.out_elems;
#pragma unroll
        for (int im = 0; im < FM; ++im)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + row;
                if (m >= a.M) continue;
                size_t ob;
                if (KIND == KIND_F) ob = (size_t)m * a.Nn;
                else {
                    int n, i, j;
                    decode_pos(m, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                    ob = ((size_t)(n * d.HB + S * i + py) * d.WB + (S * j + px)) * a.Nn;
                }
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) {
                    const int col = n0 + wn * WTN + jn * 32 + l31;
                    if (col < a.Nn) slab[ob + col] = acc[im][jn][r];
                }
            }
        return;
    }
    const bool bwd = (a.ep.kind == UAD_EPI_BWD_ACT);
    // per-lane column constants
    bool colok[FN];
    int colc[FN];
    float c_a[FN], c_b[FN];   // !bwd: bias, -- ; bwd: escale*emult, eshift
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
        const int col = n0 + wn * WTN + jn * 32 + l31;
        colok[jn] = col < a.Nn;
        colc[jn] = colok[jn] ? col : 0;
        if (!bwd) {
            c_a[jn] = a.ep.bias ? a.ep.bias[colc[jn]] : 0.f;
            c_b[jn] = 0.f;
        } else {
            c_a[jn] = a.ep.escale[colc[jn]] * a.ep.emult;
            c_b[jn] = a.ep.eshift[colc[jn]];
        }
    }
    float s1[FN], s2[FN];
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) { s1[jn] = 0.f; s2[jn] = 0.f; }

#pragma unroll
    for (int im = 0; im < FM; ++im) {
        // row bases of this fragment's 16 rows
        size_t obase[16];
        bool rowok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + row;
            rowok[r] = m < a.M;
            const int mc = rowok[r] ? m : 0;
            if (KIND == KIND_F) {
                obase[r] = (size_t)mc * a.Nn;
            } else {
                int n, i, j;
                decode_pos(mc, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                obase[r] = ((size_t)(n * d.HB + S * i + py) * d.WB + (S * j + px)) * a.Nn;
            }
        }
        if (!bwd) {
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!(rowok[r] && colok[jn])) continue;
                    const size_t off = obase[r] + colc[jn];
                    float v = acc[im][jn][r] + c_a[jn];
                    if (a.ep.mul) v *= a.ep.mul[off];
                    if (a.ep.add) v += a.ep.add[off];
                    a.Out[off] = v;
                }
            }
        } else {
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                // all 16 c_prev loads of this fragment are issued before the first one is consumed
                float cp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = rowok[r] && colok[jn];
                    cp[r] = a.ep.cprev[ok ? (obase[r] + colc[jn]) : 0];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = rowok[r] && colok[jn];
                    const float c = cp[r];
                    const float bn = fmaf(c_a[jn], c, c_b[jn]);
                    const float v = acc[im][jn][r];
                    float dbn = bn > 0.f ? v : v * a.ep.ealpha;
                    dbn = ok ? dbn : 0.f;
                    if (ok) a.Out[obase[r] + colc[jn]] = dbn * c_a[jn];
                    s1[jn] += dbn;
                    s2[jn] = fmaf(dbn, c, s2[jn]);
                }
            }
        }
    }
    if (bwd) {
        // column sums: lane halves -> waves along M -> one partial row per block tile
        float* red = smem;  // [WGM][2][BN]
        __syncthreads();
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            float t1 = s1[jn] + __shfl_xor(s1[jn], 32);
            float t2 = s2[jn] + __shfl_xor(s2[jn], 32);
            if (lh == 0) {
                red[(wm * 2 + 0) * BN + wn * WTN + jn * 32 + l31] = t1;
                red[(wm * 2 + 1) * BN + wn * WTN + jn * 32 + l31] = t2;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) t += red[(w * 2 + which) * BN + c];
            const int col = n0 + c;
            const size_t tile = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
            if (col < a.Nn) a.ep.colpart[(tile * 2 + which) * a.Nn + col] = t;
        }
    }
}

#include "uad_conv5_f32.inc"      // conv5_f_kernel / conv5_d_kernel: the exact-fp32 k5 s2 spatial kernels

// ---- generic (any KS / S / P) contraction in bf16x3 math: the fp32 tiles are split into bf16 hi | lo planes while they are stored
// to LDS (same bytes as fp32), both operands K-contiguous, so a fragment is one ds_read_b128 per plane and a K = 16 slice costs
// three v_mfma_f32_32x32x16_bf16 instead of eight fp32 MFMAs.  Identity activation-on-load only.
template <int BM, int BN, int BK, int WGM, int WGN, int KIND>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_gemm16_kernel(const ConvGemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
    static_assert(BK % 16 == 0, "BK multiple of 16");
    constexpr int NKK = BK / 16;
    constexpr int LDK = BK + 8;                        // bf16 row stride (80 B at BK = 32): 16-byte aligned rows
    constexpr int A_EL = BM * LDK, B_EL = BN * LDK;    // per plane; both operands are stored K-contiguous
    constexpr int STAGE = 2 * A_EL + 2 * B_EL;         // A hi | A lo | B hi | B lo
    __shared__ __attribute__((aligned(16))) unsigned short smem16[2 * STAGE];
    float* const smem = reinterpret_cast<float*>(smem16);   // epilogue scratch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const UadConvDesc& d = a.d;
    const int S = d.S, P = d.P, KS = d.KS;

    // ---- tap set: all KS*KS taps (F) or the taps of this output-parity class (D) ----
    int py = 0, px = 0, ky0 = 0, kx0 = 0, nty = KS, ntx = KS, dy0 = 0, dx0 = 0;
    const int nsplit = a.nsplit;
    const int split = (KIND == KIND_D) ? (int)(blockIdx.z % nsplit) : (int)blockIdx.z;
    if (KIND == KIND_D) {
        const int cz = blockIdx.z / nsplit;
        py = (S - 1) - cz / S;
        px = (S - 1) - cz % S;
        ky0 = (py + P) % S;
        kx0 = (px + P) % S;
        dy0 = (py + P) / S;
        dx0 = (px + P) / S;
        nty = ky0 < KS ? (KS - ky0 + S - 1) / S : 0;
        ntx = kx0 < KS ? (KS - kx0 + S - 1) / S : 0;
    }
    const int ntaps = nty * ntx;
    const int ncc = a.CA / BK;
    const int nk = ntaps * ncc;
    // split-K: this workgroup contracts K-steps [ks0, ks1)
    const int kper = (nk + nsplit - 1) / nsplit;
    const int ks0 = split * kper;
    const int ks1 = min(nk, ks0 + kper);

    // (no activation-on-load in this variant: the launcher only selects it for identity transforms)

    // ---- per-thread A rows (positions) ----
    constexpr int TPR = BK / 4, RPP = NT / TPR, PA = (BM + RPP - 1) / RPP;
    const int acol = (tid % TPR) * 4;
    const int arow0 = tid / TPR;
    int rbase[PA], ry[PA], rx[PA];
    bool rok[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int r = arow0 + q * RPP;
        const int m = m0 + r;
        rok[q] = (r < BM) && (m < a.M);
        int n, i, j;
        decode_pos(rok[q] ? m : 0, d.HS, d.WS, a.lhs, a.lws, n, i, j);
        if (KIND == KIND_F) {
            rbase[q] = n * d.HB * d.WB;
            ry[q] = S * i - P;
            rx[q] = S * j - P;
        } else {
            rbase[q] = n * d.HS * d.WS;
            ry[q] = i;
            rx[q] = j;
        }
    }
    const int AH = (KIND == KIND_F) ? d.HB : d.HS;
    const int AW = (KIND == KIND_F) ? d.WB : d.WS;

    // ---- per-thread B slots (predicates are loop invariant; out-of-range slots read element 0 and are zeroed) ----
    constexpr int TPRB = (KIND == KIND_F) ? (BN / 4) : (BK / 4);
    constexpr int RPB = NT / TPRB;
    constexpr int BROWS = (KIND == KIND_F) ? BK : BN;
    constexpr int PB = (BROWS + RPB - 1) / RPB;
    static_assert(KIND != KIND_F || PB == 2, "F-type: each thread owns two adjacent K rows of the weight tile");
    const int bcol = (tid % TPRB) * 4;
    const int brow0 = tid / TPRB;
    // F-type: rows (2*brow0, 2*brow0+1) so that the transposed LDS store writes one 32-bit pair per output column
    auto brow = [&](int q) { return (KIND == KIND_F) ? (brow0 * PB + q) : (brow0 + q * RPB); };
    bool bok[PB];
    int boff[PB];   // loop-invariant part of the weight offset
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int r = brow(q);
        if (KIND == KIND_F) {
            bok[q] = (r < BK) && (n0 + bcol) < a.Nn;
            boff[q] = bok[q] ? (r * d.CS + n0 + bcol) : 0;
        } else {
            bok[q] = (r < BN) && (n0 + r) < a.Nn;
            boff[q] = bok[q] ? ((n0 + r) * d.CS + bcol) : 0;
        }
    }

    float4 va[PA], vb[PB];
    unsigned okbits = 0;
    int xc0 = 0;

    // branch-free tile fetch: every load is issued unconditionally from a clamped (always valid) address
    auto load_tiles = [&](int tap, int c0) {
        int ky, kx, oy, ox;
        {
            const int ty = tap / ntx, tx = tap - ty * ntx;
            if (KIND == KIND_F) { ky = ty; kx = tx; oy = ky; ox = kx; }
            else { ky = ky0 + S * ty; kx = kx0 + S * tx; oy = dy0 - ty; ox = dx0 - tx; }
        }
        xc0 = c0;
        okbits = 0;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int y = ry[q] + oy, x = rx[q] + ox;
            const bool ok = rok[q] && (unsigned)y < (unsigned)AH && (unsigned)x < (unsigned)AW;
            const int pix = ok ? (rbase[q] + y * AW + x) : 0;
            va[q] = *reinterpret_cast<const float4*>(a.A + (size_t)pix * a.CA + c0 + acol);
            okbits |= (ok ? 1u : 0u) << q;
        }
        const int tapw = ky * KS + kx;
        const size_t wbase = (KIND == KIND_F) ? ((size_t)tapw * d.CB + c0) * d.CS : ((size_t)tapw * d.CB) * d.CS + c0;
#pragma unroll
        for (int q = 0; q < PB; ++q) vb[q] = *reinterpret_cast<const float4*>(a.W + wbase + boff[q]);
    };
    auto store_tiles = [&](int buf) {
        unsigned short* sAh = smem16 + buf * STAGE;
        unsigned short* sAl = sAh + A_EL;
        unsigned short* sBh = sAl + A_EL;
        unsigned short* sBl = sBh + B_EL;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = arow0 + q * RPP;
            // padding pixels stay exactly 0
            const float4 v = keep4((okbits >> q) & 1u, va[q]);
            uint2 hi, lo;
            split_bf16(v, hi, lo);
            if (r < BM) {
                *reinterpret_cast<uint2*>(sAh + r * LDK + acol) = hi;
                *reinterpret_cast<uint2*>(sAl + r * LDK + acol) = lo;
            }
        }
        if (KIND == KIND_F) {
            // W tile arrives [k][n]; the matrix cores want n-major rows with K contiguous: transposed store, one (k, k+1) pair per n
            uint2 h0, l0, h1, l1;
            split_bf16(keep4(bok[0], vb[0]), h0, l0);
            split_bf16(keep4(bok[PB - 1], vb[PB - 1]), h1, l1);
            const unsigned hh0[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
            const unsigned hh1[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
            const unsigned ll0[4] = {l0.x & 0xFFFFu, l0.x >> 16, l0.y & 0xFFFFu, l0.y >> 16};
            const unsigned ll1[4] = {l1.x & 0xFFFFu, l1.x >> 16, l1.y & 0xFFFFu, l1.y >> 16};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<unsigned*>(sBh + (bcol + i) * LDK + 2 * brow0) = hh0[i] | (hh1[i] << 16);
                *reinterpret_cast<unsigned*>(sBl + (bcol + i) * LDK + 2 * brow0) = ll0[i] | (ll1[i] << 16);
            }
        } else {
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int r = brow0 + q * RPB;
                uint2 hi, lo;
                split_bf16(keep4(bok[q], vb[q]), hi, lo);
                if (r < BROWS) {
                    *reinterpret_cast<uint2*>(sBh + r * LDK + bcol) = hi;
                    *reinterpret_cast<uint2*>(sBl + r * LDK + bcol) = lo;
                }
            }
        }
    };

    v16f acc[FM][FN];
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[im][jn][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;

    int tap = ks0 / ncc, cc = ks0 - tap * ncc;
    if (ks0 < ks1) {
        load_tiles(tap, cc * BK);
        store_tiles(0);
    }
    __syncthreads();

    // fragment double buffer: the ds_reads of k-slice kk+1 are in flight while the MFMAs of kk issue
    uint4 ah[2][FM], al[2][FM], bh[2][FN], bl[2][FN];
    auto load_frags = [&](int slot, const unsigned short* sAh, int kk) {
        const unsigned short* sAl = sAh + A_EL;
        const unsigned short* sBh = sAl + A_EL;
        const unsigned short* sBl = sBh + B_EL;
#pragma unroll
        for (int im = 0; im < FM; ++im) {
            const int o = (wm * WTM + im * 32 + l31) * LDK + kk * 16 + 8 * lh;
            ah[slot][im] = *reinterpret_cast<const uint4*>(sAh + o);
            al[slot][im] = *reinterpret_cast<const uint4*>(sAl + o);
        }
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            const int o = (wn * WTN + jn * 32 + l31) * LDK + kk * 16 + 8 * lh;
            bh[slot][jn] = *reinterpret_cast<const uint4*>(sBh + o);
            bl[slot][jn] = *reinterpret_cast<const uint4*>(sBl + o);
        }
    };

    for (int ks = ks0; ks < ks1; ++ks) {
        const int buf = (ks - ks0) & 1;
        int ntap = tap, ncc_ = cc + 1;
        if (ncc_ == ncc) { ncc_ = 0; ntap = tap + 1; }
        const bool more = (ks + 1 < ks1);
        const unsigned short* sA = smem16 + buf * STAGE;
        load_frags(0, sA, 0);
        if (more) load_tiles(ntap, ncc_ * BK);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < NKK) load_frags(cur ^ 1, sA, kk + 1);
#pragma unroll
            for (int im = 0; im < FM; ++im)
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) {
                    acc[im][jn] = mfma_bf16(ah[cur][im], bh[cur][jn], acc[im][jn]);
                    acc[im][jn] = mfma_bf16(ah[cur][im], bl[cur][jn], acc[im][jn]);
                    acc[im][jn] = mfma_bf16(al[cur][im], bh[cur][jn], acc[im][jn]);
                }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
        tap = ntap;
        cc = ncc_;
    }

    // ---------------------------------------------------------------- epilogue
    if (nsplit > 1) {
        // raw partial tile into this split's slab (output layout); bias / activation-backward / column sums are
        // applied by splitk_epilogue_kernel after the slabs are summed in a fixed order
        float* slab = a.Out + (size_t)split * a.out_elems;
#pragma unroll
        for (int im = 0; im < FM; ++im)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + row;
                if (m >= a.M) continue;
                size_t ob;
                if (KIND == KIND_F) ob = (size_t)m * a.Nn;
                else {
                    int n, i, j;
                    decode_pos(m, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                    ob = ((size_t)(n * d.HB + S * i + py) * d.WB + (S * j + px)) * a.Nn;
                }
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) {
                    const int col = n0 + wn * WTN + jn * 32 + l31;
                    if (col < a.Nn) slab[ob + col] = acc[im][jn][r];
                }
            }
        return;
    }
    const bool bwd = (a.ep.kind == UAD_EPI_BWD_ACT);
    // per-lane column constants
    bool colok[FN];
    int colc[FN];
    float c_a[FN], c_b[FN];   // !bwd: bias, -- ; bwd: escale*emult, eshift
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
        const int col = n0 + wn * WTN + jn * 32 + l31;
        colok[jn] = col < a.Nn;
        colc[jn] = colok[jn] ? col : 0;
        if (!bwd) {
            c_a[jn] = a.ep.bias ? a.ep.bias[colc[jn]] : 0.f;
            c_b[jn] = 0.f;
        } else {
            c_a[jn] = a.ep.escale[colc[jn]] * a.ep.emult;
            c_b[jn] = a.ep.eshift[colc[jn]];
        }
    }
    float s1[FN], s2[FN];
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) { s1[jn] = 0.f; s2[jn] = 0.f; }

#pragma unroll
    for (int im = 0; im < FM; ++im) {
        // row bases of this fragment's 16 rows
        size_t obase[16];
        bool rowok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + row;
            rowok[r] = m < a.M;
            const int mc = rowok[r] ? m : 0;
            if (KIND == KIND_F) {
                obase[r] = (size_t)mc * a.Nn;
            } else {
                int n, i, j;
                decode_pos(mc, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                obase[r] = ((size_t)(n * d.HB + S * i + py) * d.WB + (S * j + px)) * a.Nn;
            }
        }
        if (!bwd) {
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!(rowok[r] && colok[jn])) continue;
                    const size_t off = obase[r] + colc[jn];
                    float v = acc[im][jn][r] + c_a[jn];
                    if (a.ep.mul) v *= a.ep.mul[off];
                    if (a.ep.add) v += a.ep.add[off];
                    a.Out[off] = v;
                }
            }
        } else {
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                // all 16 c_prev loads of this fragment are issued before the first one is consumed
                float cp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = rowok[r] && colok[jn];
                    cp[r] = a.ep.cprev[ok ? (obase[r] + colc[jn]) : 0];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = rowok[r] && colok[jn];
                    const float c = cp[r];
                    const float bn = fmaf(c_a[jn], c, c_b[jn]);
                    const float v = acc[im][jn][r];
                    float dbn = bn > 0.f ? v : v * a.ep.ealpha;
                    dbn = ok ? dbn : 0.f;
                    if (ok) a.Out[obase[r] + colc[jn]] = dbn * c_a[jn];
                    s1[jn] += dbn;
                    s2[jn] = fmaf(dbn, c, s2[jn]);
                }
            }
        }
    }
    if (bwd) {
        // column sums: lane halves -> waves along M -> one partial row per block tile
        float* red = smem;  // [WGM][2][BN]
        __syncthreads();
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            float t1 = s1[jn] + __shfl_xor(s1[jn], 32);
            float t2 = s2[jn] + __shfl_xor(s2[jn], 32);
            if (lh == 0) {
                red[(wm * 2 + 0) * BN + wn * WTN + jn * 32 + l31] = t1;
                red[(wm * 2 + 1) * BN + wn * WTN + jn * 32 + l31] = t2;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) t += red[(w * 2 + which) * BN + c];
            const int col = n0 + c;
            const size_t tile = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
            if (col < a.Nn) a.ep.colpart[(tile * 2 + which) * a.Nn + col] = t;
        }
    }
}


// KIND_F: halo (2TH+3)x(2TW+3), stride-2 gather;  KIND_D: halo (TH+2)x(TW+2), four output-parity classes
template <int TH, int TW, int CK, int WGM, int WGN, int KIND>
__global__ void __launch_bounds__(64 * WGM * WGN) conv5_bf16_kernel(const ConvGemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int IH = (KIND == KIND_F) ? 2 * TH + 3 : TH + 2, IW = (KIND == KIND_F) ? 2 * TW + 3 : TW + 2;
    constexpr int LDH = CK + 8;                 // ushorts per pixel row (+16 B pad: odd number of 16-B slots)
    constexpr int NKS = CK / 16, CQ = CK / 4;
    constexpr int BN = 32 * WGN;
    constexpr int NACC = (KIND == KIND_F) ? 3 : 8;
    static_assert(TH * TW == 32 * WGM, "one 32-row fragment per wave along M");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned short* sHi = reinterpret_cast<unsigned short*>(dsm);
    unsigned short* sLo = sHi + IH * IW * LDH;
    float* s_xf = reinterpret_cast<float*>(sLo + IH * IW * LDH);
    float* s_red = s_xf + 2 * XF_LDS_CH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const UadConvDesc& d = a.d;
    const int tilesx = d.WS / TW;
    const int ty0 = (blockIdx.x / tilesx) * TH, tx0 = (blockIdx.x % tilesx) * TW;
    const int n = blockIdx.y;
    const int nsplit = a.nsplit;
    const int n0 = (blockIdx.z / nsplit) * BN;
    const int split = blockIdx.z % nsplit;
    const int CA = a.CA, Nn = a.Nn;   // contraction / output channels
    const int AH = (KIND == KIND_F) ? d.HB : d.HS, AW = (KIND == KIND_F) ? d.WB : d.WS;

    const bool xf = a.xf.scale != nullptr;
    if (xf)
        for (int c = tid; c < CA; c += NT) {
            s_xf[c] = a.xf.scale[c] * a.xf.mult;
            s_xf[XF_LDS_CH + c] = a.xf.shift[c];
        }

    const int m = wm * 32 + l31;
    const int pty = m / TW, ptx = m % TW;
    const int aoff = (KIND == KIND_F) ? ((2 * pty) * IW + 2 * ptx) * LDH + 8 * lh : ((pty + 1) * IW + ptx + 1) * LDH + 8 * lh;
    const int col = n0 + wn * 32 + l31;
    const bool colok = col < Nn;
    const int colc = colok ? col : 0;
    // packed planes [tap][k/8][n][8] bf16: one uint4 per (k-octet, n)
    const uint4* wq = reinterpret_cast<const uint4*>(a.Wp16) + (size_t)lh * Nn + colc;
    const size_t plane_q = (size_t)a.w16_plane / 8;

    BFrag16<NKS> b0, b1;
    auto loadB = [&](BFrag16<NKS>& b, int tap, int c0) {
        const uint4* w = wq + ((size_t)tap * (CA / 8) + c0 / 8) * Nn;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            b.hi[j] = w[(size_t)(2 * j) * Nn];
            b.lo[j] = w[(size_t)(2 * j) * Nn + plane_q];
        }
    };

    v16f acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int cper = (CA / CK + nsplit - 1) / nsplit;
    const int ch0 = split * cper;
    const int nchunks = min(CA / CK, ch0 + cper);
    const int gy0 = (KIND == KIND_F) ? 2 * ty0 - 1 : ty0 - 1, gx0 = (KIND == KIND_F) ? 2 * tx0 - 1 : tx0 - 1;
    const float* inb = a.A + (size_t)n * AH * AW * CA;
    loadB(b0, 0, ch0 * CK);
    __syncthreads();

    for (int ch = ch0; ch < nchunks; ++ch) {
        const int c0 = ch * CK;
        if (ch > ch0) __syncthreads();
        constexpr int TOT = IH * IW * CQ;
        constexpr int BATCH = (KIND == KIND_F) ? 6 : 4;
        for (int f0 = tid; f0 < TOT; f0 += NT * BATCH) {
            float4 v[BATCH];
            bool ok[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + u * NT;
                const int pix = f / CQ, cq = f % CQ;
                const int iy = pix / IW, ix = pix % IW;
                const int gy = gy0 + iy, gx = gx0 + ix;
                ok[u] = (f < TOT) && (unsigned)gy < (unsigned)AH && (unsigned)gx < (unsigned)AW;
                const int gp = ok[u] ? (gy * AW + gx) : 0;
                v[u] = *reinterpret_cast<const float4*>(inb + (size_t)gp * CA + c0 + (ok[u] ? cq * 4 : 0));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + u * NT;
                if (f >= TOT) continue;
                const int pix = f / CQ, cq = f % CQ;
                float4 t = v[u];
                if (xf) {
                    const float4 sc = *reinterpret_cast<const float4*>(s_xf + c0 + cq * 4);
                    const float4 sh = *reinterpret_cast<const float4*>(s_xf + XF_LDS_CH + c0 + cq * 4);
                    t = xform4(t, sc, sh, a.xf.alpha);
                }
                t = keep4(ok[u], t);
                uint2 hi, lo;
                split_bf16(t, hi, lo);
                *reinterpret_cast<uint2*>(sHi + pix * LDH + cq * 4) = hi;
                *reinterpret_cast<uint2*>(sLo + pix * LDH + cq * 4) = lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            const int ky = tap / 5, kx = tap % 5;
            int toff, cls = 0;
            if (KIND == KIND_F) {
                toff = (ky * IW + kx) * LDH;
            } else {
                const int py = (ky + 1) & 1, px = (kx + 1) & 1;
                const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
                toff = (dy * IW + dx) * LDH;
                cls = py * 2 + px;
            }
            BFrag16<NKS>& cur = (tap & 1) ? b1 : b0;
            BFrag16<NKS>& nxt = (tap & 1) ? b0 : b1;
            if (tap < 24) loadB(nxt, tap + 1, c0);
            else if (ch + 1 < nchunks) loadB(nxt, 0, c0 + CK);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NKS; ++j) {
                const uint4 ah = *reinterpret_cast<const uint4*>(sHi + aoff + toff + 16 * j);
                const uint4 al = *reinterpret_cast<const uint4*>(sLo + aoff + toff + 16 * j);
                if (KIND == KIND_F) {
                    acc[0] = mfma_bf16(ah, cur.hi[j], acc[0]);
                    acc[1] = mfma_bf16(ah, cur.lo[j], acc[1]);
                    acc[2] = mfma_bf16(al, cur.hi[j], acc[2]);
                } else {
                    acc[2 * cls] = mfma_bf16(ah, cur.hi[j], acc[2 * cls]);
                    acc[2 * cls + 1] = mfma_bf16(ah, cur.lo[j], acc[2 * cls + 1]);
                    acc[2 * cls + 1] = mfma_bf16(al, cur.hi[j], acc[2 * cls + 1]);
                }
            }
        }
        b0 = b1;
    }

    // ---- epilogue (identical to the fp32 spatial kernels: the C layout of the MFMA is dtype independent) ----
    const bool bwd = (a.ep.kind == UAD_EPI_BWD_ACT);
    float c_a, c_b = 0.f;
    if (!bwd) c_a = a.ep.bias ? a.ep.bias[colc] : 0.f;
    else { c_a = a.ep.escale[colc] * a.ep.emult; c_b = a.ep.eshift[colc]; }
    float s1 = 0.f, s2 = 0.f;
    constexpr int NCLS = (KIND == KIND_F) ? 1 : 4;
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls) {
        v16f o;
        if (KIND == KIND_F) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = acc[0][r] + (acc[1][r] + acc[2][r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = acc[2 * cls][r] + acc[2 * cls + 1][r];
        }
        size_t obase[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (KIND == KIND_F) {
                obase[r] = ((size_t)(n * d.HS + ty0 + mm / TW) * d.WS + tx0 + mm % TW) * Nn;
            } else {
                const int Y = 2 * (ty0 + mm / TW) + (cls >> 1), X = 2 * (tx0 + mm % TW) + (cls & 1);
                obase[r] = ((size_t)(n * d.HB + Y) * d.WB + X) * Nn;
            }
        }
        if (nsplit > 1) {
            float* slab = a.Out + (size_t)split * a.out_elems;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (colok) slab[obase[r] + colc] = o[r];
        } else {
            epilogue_frag<0>(a, o, obase, colok, colc, c_a, c_b, s1, s2);
        }
    }
    if (nsplit > 1) return;
    if (bwd) {
        const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
        if (lh == 0) {
            s_red[(wm * 2 + 0) * BN + wn * 32 + l31] = t1;
            s_red[(wm * 2 + 1) * BN + wn * 32 + l31] = t2;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) t += s_red[(w * 2 + which) * BN + c];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            if (n0 + c < Nn) a.ep.colpart[(tile * 2 + which) * Nn + n0 + c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// D-kind, class-sequential form (bf16x3).  Every tap of a k5 s2 transposed contraction feeds exactly ONE of the four
// output-parity classes, so the classes share no operand: the workgroup stages its WHOLE channel range (CST channels of the
// (TH+2)x(TW+2) halo) once, then walks the classes one after the other -- 3 accumulator fragments live instead of 8
// (340 -> ~150 registers: two-three waves per SIMD instead of one), and each class's epilogue stores overlap the next
// class's MFMAs.  Requires CA == CST * nsplit.
// ------------------------------------------------------------------------------------------------

template <int TH, int TW, int CST, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN) __attribute__((amdgpu_waves_per_eu(2))) conv5_d16_kernel(const ConvGemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int IH = TH + 2, IW = TW + 2;
    constexpr int LDH = CST + 8;                // ushorts per pixel row (+16 B pad)
    constexpr int NCH = CST / 32, CQ = CST / 4;
    constexpr int BN = 32 * WGN;
    static_assert(TH * TW == 32 * WGM, "one 32-row fragment per wave along M");
    static_assert(CST % 32 == 0, "32-channel weight-fragment units");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned short* sHi = reinterpret_cast<unsigned short*>(dsm);
    unsigned short* sLo = sHi + IH * IW * LDH;
    float* s_xf = reinterpret_cast<float*>(sLo + IH * IW * LDH);
    float* s_red = s_xf + 2 * XF_LDS_CH;
    float* s_epi = s_red + WGM * 2 * BN;        // per-wave 32 x EPI_LD transpose tile of the epilogue
    float* s_fx = s_epi + WGM * WGN * 32 * 36;  // EPI_FINAL: target / reconstruction / L1 tiles of the 2TH x 2TW output block
    float* s_fo = s_fx + 4 * TH * TW;
    float* s_fl = s_fo + 4 * TH * TW;
    unsigned* s_fb = reinterpret_cast<unsigned*>(s_fl + 4 * TH * TW);   // EPI_FINAL + fin_bits: pattern word / d objective / d x_hat per pixel
    float* s_fg = reinterpret_cast<float*>(s_fb + 4 * TH * TW);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const UadConvDesc& d = a.d;
    const int tilesx = d.WS / TW;
    const int ty0 = (blockIdx.x / tilesx) * TH, tx0 = (blockIdx.x % tilesx) * TW;
    const int n = blockIdx.y;
    const int nsplit = a.nsplit;
    const int n0 = (blockIdx.z / nsplit) * BN;
    const int split = blockIdx.z % nsplit;
    const int CA = a.CA, Nn = a.Nn;
    const int cbase = split * CST;
    const int AH = d.HS, AW = d.WS;

    const bool xf = a.xf.scale != nullptr;
    if (xf)
        for (int c = tid; c < CST; c += NT) {
            s_xf[c] = a.xf.scale[cbase + c] * a.xf.mult;
            s_xf[XF_LDS_CH + c] = a.xf.shift[cbase + c];
        }

    if (a.ep.kind == UAD_EPI_FINAL)
        for (int idx = tid; idx < 4 * TH * TW; idx += NT) {
            const int yl = idx / (2 * TW), xl = idx % (2 * TW);
            s_fx[idx] = a.ep.fin_x[((size_t)n * d.HB + 2 * ty0 + yl) * d.WB + 2 * tx0 + xl];
        }

    const int m = wm * 32 + l31;
    const int pty = m / TW, ptx = m % TW;
    const int aoff = ((pty + 1) * IW + ptx + 1) * LDH + 8 * lh;
    const int col = n0 + wn * 32 + l31;
    const bool colok = col < Nn;
    const int colc = colok ? col : 0;
    const size_t plane_q = (size_t)a.w16_plane / 8;

    // Weight fragments: global -> VGPR ring of four, three units (one unit = one tap x 32 channels) ahead of their use: the
    // loads are L2 hits (a layer's planes exceed the 32 KB L1), ~500+ cycles against 192 cycles of MFMA work per unit.
    // (Sharing them through LDS was measured slower: it doubles the LDS read traffic, which then bounds the loop.)
    // Activation fragments: LDS -> VGPR, double-buffered one unit ahead, so the MFMAs of a unit never wait for LDS.
    constexpr int NUNIT = 25 * NCH;
    const uint4* wq = reinterpret_cast<const uint4*>(a.Wp16) + (size_t)lh * Nn + colc;
    BFrag16<2> b0, b1, b2, b3;
    // wave-uniform base (SGPR pair, scalar arithmetic) + one 32-bit per-lane offset: no 64-bit VALU add per load
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Wp16), 0, 0xffffffff, 0x00020000);
    const unsigned lb = (unsigned)(lh * Nn + colc) * 16u;      // byte offsets, 32 bits: the packed weights of a layer are < 4 GB
    const unsigned plane_b = (unsigned)plane_q * 16u;
    auto loadB = [&](BFrag16<2>& b, int unit) __attribute__((always_inline)) {
        const int tap = kTapOrderD.tap[unit / NCH], c0 = cbase + (unit % NCH) * 32;
        const unsigned u = (unsigned)((tap * (CA / 8) + c0 / 8) * Nn) * 16u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b.hi[j] = buf_load16(wrs, lb, u + (unsigned)(2 * j * Nn) * 16u);
            b.lo[j] = buf_load16(wrs, lb, u + (unsigned)(2 * j * Nn) * 16u + plane_b);
        }
    };
    auto loadA = [&](BFrag16<2>& f, int unit) __attribute__((always_inline)) {
        const int tap = kTapOrderD.tap[unit / NCH], kc = unit % NCH;
        const int ky = tap / 5, kx = tap % 5;
        const int py = (ky + 1) & 1, px = (kx + 1) & 1;
        const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
        const int toff = (dy * IW + dx) * LDH;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f.hi[j] = *reinterpret_cast<const uint4*>(sHi + aoff + toff + kc * 32 + 16 * j);
            f.lo[j] = *reinterpret_cast<const uint4*>(sLo + aoff + toff + kc * 32 + 16 * j);
        }
    };
    loadB(b0, 0);
    if (1 < NUNIT) loadB(b1, 1);
    if (2 < NUNIT) loadB(b2, 2);
    if (xf) __syncthreads();

    // ---- stage the whole halo tile of this workgroup's channel range: global fp32 -> activation -> bf16 hi|lo planes ----
    const int gy0 = ty0 - 1, gx0 = tx0 - 1;
    const float* inb = a.A + (size_t)n * AH * AW * CA + cbase;
    {
        constexpr int TOT = IH * IW * CQ;
        constexpr int BATCH = 4;
        for (int f0 = tid; f0 < TOT; f0 += NT * BATCH) {
            float4 v[BATCH];
            bool ok[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + u * NT;
                const int pix = f / CQ, cq = f % CQ;
                const int iy = pix / IW, ix = pix % IW;
                const int gy = gy0 + iy, gx = gx0 + ix;
                ok[u] = (f < TOT) && (unsigned)gy < (unsigned)AH && (unsigned)gx < (unsigned)AW;
                const int gp = ok[u] ? (gy * AW + gx) : 0;
                v[u] = *reinterpret_cast<const float4*>(inb + (size_t)gp * CA + (ok[u] ? cq * 4 : 0));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + u * NT;
                if (f >= TOT) continue;
                const int pix = f / CQ, cq = f % CQ;
                float4 t = v[u];
                if (xf) {
                    const float4 sc = *reinterpret_cast<const float4*>(s_xf + cq * 4);
                    const float4 sh = *reinterpret_cast<const float4*>(s_xf + XF_LDS_CH + cq * 4);
                    t = xform4(t, sc, sh, a.xf.alpha);
                }
                t = keep4(ok[u], t);
                uint2 hi, lo;
                split_bf16(t, hi, lo);
                *reinterpret_cast<uint2*>(sHi + pix * LDH + cq * 4) = hi;
                *reinterpret_cast<uint2*>(sLo + pix * LDH + cq * 4) = lo;
            }
        }
    }
    __syncthreads();

    // Epilogue: the 32x32 C fragment (lane = column) is transposed through a wave-private LDS tile so that every lane owns
    // 4 consecutive channels of 4 rows: 4 x 16-byte stores (and cprev loads) per class instead of 16 x 4-byte ones -- the
    // scattered dword form spent ~190 cycles per store instruction in the address path (3000 of 10000 cycles per class).
    constexpr int EPI_LD = 36;
    const bool bwd = (a.ep.kind == UAD_EPI_BWD_ACT);
    float* etile = s_epi + wave * 32 * EPI_LD;
    const int ec4 = (lane & 7) * 4, erow = lane >> 3;
    const int ecol = n0 + wn * 32 + ec4;                 // Nn % 32 == 0 on this path: always in range
    float4 e_a = make_float4(0.f, 0.f, 0.f, 0.f), e_b = e_a;
    if (!bwd) { if (a.ep.bias) e_a = *reinterpret_cast<const float4*>(a.ep.bias + ecol); }
    else {
        e_a = *reinterpret_cast<const float4*>(a.ep.escale + ecol);
        e_a.x *= a.ep.emult; e_a.y *= a.ep.emult; e_a.z *= a.ep.emult; e_a.w *= a.ep.emult;
        e_b = *reinterpret_cast<const float4*>(a.ep.eshift + ecol);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // EPI_FINAL: e_a := conv bias, f_sc/f_sh := this layer's BN, f_w := final 1x1 kernel; per-lane running sums of the
    // final conv's kernel gradient (dwf), the loss (rec) and the final bias gradient (dbf)
    const bool fin = (a.ep.kind == UAD_EPI_FINAL);
    float4 f_sc = make_float4(0.f, 0.f, 0.f, 0.f), f_sh = f_sc, f_w = f_sc;
    float f_bf = 0.f, rec = 0.f, dbf = 0.f;
    float dwf[4] = {0.f, 0.f, 0.f, 0.f};
    if (fin) {
        f_sc = *reinterpret_cast<const float4*>(a.ep.escale + ecol);
        f_sc.x *= a.ep.emult; f_sc.y *= a.ep.emult; f_sc.z *= a.ep.emult; f_sc.w *= a.ep.emult;
        f_sh = *reinterpret_cast<const float4*>(a.ep.eshift + ecol);
        f_w = *reinterpret_cast<const float4*>(a.ep.fin_wf + ecol);
        f_bf = a.ep.fin_bf[0];
    }
    // reduce = false: the accumulators of one parity class; reduce = true (split-K, last workgroup of the tile): the class's sum over the slabs
    auto class_epilogue = [&](const v16f& o, int py, int px, const bool reduce) __attribute__((always_inline)) {
        if (!reduce) {
#pragma unroll
            for (int r = 0; r < 16; ++r) etile[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_LD + l31] = o[r];
        }
        __builtin_amdgcn_wave_barrier();
        float4 v[4];
        size_t off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = erow + 8 * k;
            v[k] = reduce ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(etile + row * EPI_LD + ec4);
            const int mm = wm * 32 + row;
            const int Y = 2 * (ty0 + mm / TW) + py, X = 2 * (tx0 + mm % TW) + px;
            off[k] = ((size_t)(n * d.HB + Y) * d.WB + X) * Nn + ecol;
        }
        __builtin_amdgcn_wave_barrier();
        float* outp = a.Out;
        if (nsplit > 1 && reduce) {
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, 0xffffffff, 0x00020000);
            for (int sp = 0; sp < nsplit; sp += 2) {          // split order, two slabs (8 loads) in flight
                float4 t[4][2];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int h = 0; h < 2; ++h) t[k][h] = sk_load16(srs, (unsigned)(((size_t)(sp + h) * a.out_elems + off[k]) * 4));
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { v[k].x += t[k][h].x; v[k].y += t[k][h].y; v[k].z += t[k][h].z; v[k].w += t[k][h].w; }
            }
            outp = a.out_final;
        }
        if (nsplit > 1 && !reduce) {
            if (a.sk_counter) {
                const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, 0xffffffff, 0x00020000);
#pragma unroll
                for (int k = 0; k < 4; ++k) sk_store16(srs, (unsigned)(((size_t)split * a.out_elems + off[k]) * 4), v[k]);
            } else {
                float* slab = a.Out + (size_t)split * a.out_elems;
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(slab + off[k]) = v[k];
            }
        } else if (fin) {
            // last decoder block: BN + LeakyReLU, final 1x1 conv (C -> 1, reduced over the 8 lanes that share a pixel), L1 loss
            // and, if wanted, its gradient back to this layer's pre-BN output (models/customlayers.py:35-37, trainers/VAE.py:36-40)
            int lidx[4];
            float xin[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int mm = wm * 32 + erow + 8 * k;
                lidx[k] = (2 * (mm / TW) + py) * (2 * TW) + 2 * (mm % TW) + px;
                xin[k] = s_fx[lidx[k]];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float cc[4] = {v[k].x + e_a.x, v[k].y + e_a.y, v[k].z + e_a.z, v[k].w + e_a.w};
                const float scv[4] = {f_sc.x, f_sc.y, f_sc.z, f_sc.w}, shv[4] = {f_sh.x, f_sh.y, f_sh.z, f_sh.w};
                const float wv[4] = {f_w.x, f_w.y, f_w.z, f_w.w};
                float bn[4], av[4];
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bn[e] = fmaf(cc[e], scv[e], shv[e]);
                    av[e] = bn[e] > 0.f ? bn[e] : bn[e] * a.ep.ealpha;
                    dot = fmaf(av[e], wv[e], dot);
                }
                dot = sum8_dpp(dot);
                const float xh = dot + f_bf;
                const float diff = xh - xin[k];
                // all 8 lanes of the pixel hold the same xh / diff: lanes 0..3 of the group each store one of the four per-pixel values
                // (x_hat, |diff|, pattern word, sign), the running sums are kept in every lane and read from lane 0 of the group
                rec += fabsf(diff);
                if (a.Out) *reinterpret_cast<float4*>(a.Out + off[k]) = make_float4(cc[0], cc[1], cc[2], cc[3]);
                unsigned pw = 0u;
                float sg = 0.f;
                if (a.ep.fin_dc || a.ep.fin_bits) {
                    const float sgn = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * a.ep.fin_inv_batch;
                    float dc[4];
                    unsigned nib = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float da = sgn * wv[e];
                        const bool pos = bn[e] > 0.f;
                        const float dbn = pos ? da : da * a.ep.ealpha;
                        dc[e] = dbn * scv[e];
                        dwf[e] = fmaf(sgn, av[e], dwf[e]);
                        s1[e] += dbn;
                        s2[e] = fmaf(dbn, cc[e], s2[e]);
                        nib |= (pos ? 1u : 0u) << e;
                    }
                    if (a.ep.fin_dc) *reinterpret_cast<float4*>(a.ep.fin_dc + off[k]) = make_float4(dc[0], dc[1], dc[2], dc[3]);
                    // d loss / d c of this pixel = sgn * w_f[ch] * (bit ? 1 : alpha) * scale[ch]: one word + one float instead of 32 floats
                    pw = or8_dpp(nib << ecol);
                    sg = sgn;
                    dbf += sgn;
                }
                {
                    const int role = lane & 3;
                    const float val = role == 0 ? xh : role == 1 ? fabsf(diff) : role == 2 ? __uint_as_float(pw) : sg;
                    if ((lane & 4) == 0) s_fo[role * (4 * TH * TW) + lidx[k]] = val;      // s_fo | s_fl | s_fb | s_fg are adjacent
                }
            }
        } else if (!bwd) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 t = v[k];
                t.x += e_a.x; t.y += e_a.y; t.z += e_a.z; t.w += e_a.w;
                if (a.ep.mul) { const float4 q = *reinterpret_cast<const float4*>(a.ep.mul + off[k]); t.x *= q.x; t.y *= q.y; t.z *= q.z; t.w *= q.w; }
                if (a.ep.add) { const float4 q = *reinterpret_cast<const float4*>(a.ep.add + off[k]); t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
                *reinterpret_cast<float4*>(outp + off[k]) = t;
            }
        } else {
            float4 cp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cp[k] = *reinterpret_cast<const float4*>(a.ep.cprev + off[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, cc[4] = {cp[k].x, cp[k].y, cp[k].z, cp[k].w};
                const float aa[4] = {e_a.x, e_a.y, e_a.z, e_a.w}, bb[4] = {e_b.x, e_b.y, e_b.z, e_b.w};
                float oo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bn = fmaf(aa[e], cc[e], bb[e]);
                    const float dbn = bn > 0.f ? vv[e] : vv[e] * a.ep.ealpha;
                    oo[e] = dbn * aa[e];
                    s1[e] += dbn;
                    s2[e] = fmaf(dbn, cc[e], s2[e]);
                }
                *reinterpret_cast<float4*>(outp + off[k]) = make_float4(oo[0], oo[1], oo[2], oo[3]);
            }
        }
    };

    // The unit sequence is expanded at compile time (static_for, not `#pragma unroll`: the optimizer declined to unroll the loop form
    // and then selected tap / class / accumulator-reset at run time -- 48 v_cndmask per unit of 6 MFMAs).
    v16f acc0, acc1, acc2;
    BFrag16<2> a0, a1;
    loadA(a0, 0);
    auto unit = [&](const BFrag16<2>& bc, BFrag16<2>& bpf, const BFrag16<2>& ac, BFrag16<2>& an, auto S) __attribute__((always_inline)) {
        constexpr int sidx = decltype(S)::value;
        constexpr int t = sidx / NCH, kc = sidx % NCH;
        constexpr int tap = kTapOrderD.tap[t];
        constexpr int ky = tap / 5, kx = tap % 5;
        constexpr int py = (ky + 1) & 1, px = (kx + 1) & 1;
        constexpr int cls = py * 2 + px;
        constexpr bool first = (kc == 0 && t == kTapOrderD.start[cls]);
        constexpr bool last = (kc == NCH - 1 && t + 1 == kTapOrderD.start[cls + 1]);
        if constexpr (sidx + 3 < NUNIT) loadB(bpf, sidx + 3);
        if constexpr (sidx + 1 < NUNIT) loadA(an, sidx + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (first) {
            const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc0 = mfma_bf16(ac.hi[0], bc.hi[0], z);
            acc1 = mfma_bf16(ac.hi[0], bc.lo[0], z);
            acc2 = mfma_bf16(ac.lo[0], bc.hi[0], z);
        } else {
            acc0 = mfma_bf16(ac.hi[0], bc.hi[0], acc0);
            acc1 = mfma_bf16(ac.hi[0], bc.lo[0], acc1);
            acc2 = mfma_bf16(ac.lo[0], bc.hi[0], acc2);
        }
        acc0 = mfma_bf16(ac.hi[1], bc.hi[1], acc0);
        acc1 = mfma_bf16(ac.hi[1], bc.lo[1], acc1);
        acc2 = mfma_bf16(ac.lo[1], bc.hi[1], acc2);
        if constexpr (last) {
            // ---- epilogue of this parity class ----
            v16f o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = acc0[r] + (acc1[r] + acc2[r]);
            class_epilogue(o, py, px, false);
        }
    };
    static_for<0, (NUNIT + 3) / 4>([&](auto G) __attribute__((always_inline)) {
        constexpr int g4 = decltype(G)::value;
        if constexpr (4 * g4 + 0 < NUNIT) unit(b0, b3, a0, a1, std::integral_constant<int, 4 * g4 + 0>{});
        if constexpr (4 * g4 + 1 < NUNIT) unit(b1, b0, a1, a0, std::integral_constant<int, 4 * g4 + 1>{});
        if constexpr (4 * g4 + 2 < NUNIT) unit(b2, b1, a0, a1, std::integral_constant<int, 4 * g4 + 2>{});
        if constexpr (4 * g4 + 3 < NUNIT) unit(b3, b2, a1, a0, std::integral_constant<int, 4 * g4 + 3>{});
    });
    if (fin) {
        // workgroup partials in the final kernel's layout: red_partial[tile][3C+1] = {dwf[C], S1[C], S2[C], dbf}, rec_partial[tile]
        __syncthreads();                        // every wave is done with its transpose tile (s_epi is reused below)
        for (int idx = tid; idx < 4 * TH * TW; idx += NT) {
            const int yl = idx / (2 * TW), xl = idx % (2 * TW);
            const size_t pix = ((size_t)n * d.HB + 2 * ty0 + yl) * d.WB + 2 * tx0 + xl;
            a.ep.fin_xhat[pix] = s_fo[idx];
            if (a.ep.fin_l1) a.ep.fin_l1[pix] = s_fl[idx];
            if (a.ep.fin_bits) { a.ep.fin_bits[pix] = s_fb[idx]; a.ep.fin_dxhat[pix] = s_fg[idx]; }
        }
        float* fr = s_epi;                      // [waves][3*BN + 2]
        constexpr int FL = 3 * BN + 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t0 = dwf[e], t1 = s1[e], t2 = s2[e];
            t0 += __shfl_xor(t0, 8); t0 += __shfl_xor(t0, 16); t0 += __shfl_xor(t0, 32);
            t1 += __shfl_xor(t1, 8); t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
            t2 += __shfl_xor(t2, 8); t2 += __shfl_xor(t2, 16); t2 += __shfl_xor(t2, 32);
            if (lane < 8) {
                fr[wave * FL + 0 * BN + wn * 32 + ec4 + e] = t0;
                fr[wave * FL + 1 * BN + wn * 32 + ec4 + e] = t1;
                fr[wave * FL + 2 * BN + wn * 32 + ec4 + e] = t2;
            }
        }
        float r0 = rec, r1 = dbf;
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { r0 += __shfl_xor(r0, o); r1 += __shfl_xor(r1, o); }
        if (lane == 0) { fr[wave * FL + 3 * BN] = r1; fr[wave * FL + 3 * BN + 1] = r0; }
        __syncthreads();
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (tid < FL) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WGM * WGN; ++w) t += fr[w * FL + tid];
            if (tid == 3 * BN + 1) a.ep.fin_rec_partial[tile] = t;
            else if (a.ep.fin_dc || a.ep.fin_bits) a.ep.fin_red_partial[tile * (3 * BN + 1) + tid] = t;
        }
        return;
    }
    if (nsplit > 1) {
        if (!a.sk_counter) return;            // splitk_epilogue_kernel finishes the job
        const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) * (gridDim.z / nsplit) + blockIdx.z / nsplit;
        if (!sk_last_arriver(a.sk_counter + slot, nsplit, reinterpret_cast<int*>(s_red), tid)) return;
        const v16f none = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        class_epilogue(none, 0, 0, true); class_epilogue(none, 0, 1, true); class_epilogue(none, 1, 0, true); class_epilogue(none, 1, 1, true);
    }
    if (bwd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t1 = s1[e], t2 = s2[e];
            t1 += __shfl_xor(t1, 8); t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
            t2 += __shfl_xor(t2, 8); t2 += __shfl_xor(t2, 16); t2 += __shfl_xor(t2, 32);
            if (lane < 8) {
                s_red[(wm * 2 + 0) * BN + wn * 32 + ec4 + e] = t1;
                s_red[(wm * 2 + 1) * BN + wn * 32 + ec4 + e] = t2;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) t += s_red[(w * 2 + which) * BN + c];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            if (n0 + c < Nn) a.ep.colpart[(tile * 2 + which) * Nn + n0 + c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F-kind (gather) bf16x3 kernel with the same pipeline as conv5_d16_kernel: weight fragments three taps ahead in a VGPR ring,
// activation fragments double-buffered out of LDS one tap ahead, transposed 16-byte epilogue.  The (2TH+3)x(2TW+3) halo is
// too large to hold every channel at once, so channels are walked in CK-wide chunks (restaged per chunk, accumulators live
// across chunks); the weight ring runs through the chunk boundary.
// ------------------------------------------------------------------------------------------------
// Round 6: the staging phases on the same diet as conv5_w_bf16_tr_kernel -- the tile is fixed for a workgroup, so the byte offset of each of a
// thread's elements (out-of-image elements: 2^31 = the descriptor's size, the range check returns zeros), the padding mask and the LDS store
// addresses are computed ONCE; a chunk adds a wave-uniform channel offset.  Activation-on-load or not is a template parameter (XF).
template <int TH, int TW, int CK, int WGM, int WGN, int FB, bool XF, int NPL = 2>      // FB: 0 plain | 1 final-backward on load from c | 2 ... from the pattern bits; NPL: bf16 planes per operand (3 = bf16x6)
__global__ void __launch_bounds__(64 * WGM * WGN) __attribute__((amdgpu_waves_per_eu(2))) conv5_f16_kernel(const ConvGemmArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int IH = 2 * TH + 3, IW = 2 * TW + 3;
    constexpr int LDH = CK + 8;
    constexpr int NKS = CK / 16, CQ = CK / 4;
    constexpr int BN = 32 * WGN;
    static_assert(TH * TW == 32 * WGM, "one 32-row fragment per wave along M");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    constexpr int PLANE = IH * IW * LDH;
    unsigned short* sPl = reinterpret_cast<unsigned short*>(dsm);      // NPL planes of IH x IW x LDH
    float* s_xf = reinterpret_cast<float*>(sPl + NPL * PLANE);
    float* s_red = s_xf + 3 * XF_LDS_CH;      // s_xf: scale | shift | final-conv kernel (fb mode)
    float* s_epi = reinterpret_cast<float*>(dsm);      // aliases the activation tile (dead after the last chunk)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const UadConvDesc& d = a.d;
    const int tilesx = d.WS / TW;
    const int ty0 = (blockIdx.x / tilesx) * TH, tx0 = (blockIdx.x % tilesx) * TW;
    const int n = blockIdx.y;
    const int nsplit = a.nsplit;
    const int n0 = (blockIdx.z / nsplit) * BN;
    const int split = blockIdx.z % nsplit;
    const int CA = a.CA, Nn = a.Nn;
    const int AH = d.HB, AW = d.WB;

    constexpr bool xf = XF;
    constexpr bool fb = FB == 1 || FB == 2;  // final-backward on load (UadXform::fb_*): its own instantiation, the extra
                                             // prefetch registers would otherwise spill the 64-column variant
    constexpr bool fbb = FB == 2;            // ... from one pattern word per pixel instead of the 32 pre-BN values (UadXform::fb_bits)
    static_assert(!(fb && !XF), "the final-backward forms are instantiated with XF = true (they read the scale table)");
    if (xf)
        for (int c = tid; c < CA; c += NT) {
            s_xf[c] = a.xf.scale[c] * a.xf.mult;
            s_xf[XF_LDS_CH + c] = a.xf.shift[c];
            if (fb) s_xf[2 * XF_LDS_CH + c] = a.xf.fb_wf[c];
        }
    const int m = wm * 32 + l31;
    const int pty = m / TW, ptx = m % TW;
    const int aoff = ((2 * pty) * IW + 2 * ptx) * LDH + 8 * lh;
    const int col = n0 + wn * 32 + l31;
    const int colc = col < Nn ? col : 0;
    const uint4* wq = reinterpret_cast<const uint4*>(a.Wp16) + (size_t)lh * Nn + colc;
    const size_t plane_q = (size_t)a.w16_plane / 8;

    const int cper = (CA / CK + nsplit - 1) / nsplit;
    const int ch0 = split * cper;
    const int nchunks = min(CA / CK, ch0 + cper);

    KFrag<NKS, NPL> b0, b1, b2, b3, a0, a1;
    // wave-uniform base (SGPR pair, scalar arithmetic) + one 32-bit per-lane offset: no 64-bit VALU add per load
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Wp16), 0, 0xffffffff, 0x00020000);
    const unsigned lb = (unsigned)(wq - reinterpret_cast<const uint4*>(a.Wp16)) * 16u;      // byte offsets, 32 bits: the packed weights of a layer are < 4 GB
    const unsigned plane_b = (unsigned)plane_q * 16u;
    auto loadB = [&](KFrag<NKS, NPL>& b, int tap, int c0) {
        const unsigned u = (unsigned)((tap * (CA / 8) + c0 / 8) * Nn) * 16u;
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) b.p[p][j] = buf_load16(wrs, lb, u + (unsigned)(2 * j * Nn) * 16u + (unsigned)p * plane_b);
    };

    auto loadA = [&](KFrag<NKS, NPL>& f, int tap) {
        const int toff = ((tap / 5) * IW + (tap % 5)) * LDH;
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) f.p[p][j] = *reinterpret_cast<const uint4*>(sPl + p * PLANE + aoff + toff + 16 * j);
    };
    loadB(b0, 0, ch0 * CK);
    loadB(b1, 1, ch0 * CK);
    loadB(b2, 2, ch0 * CK);

    v16f acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    const int gy0 = 2 * ty0 - 1, gx0 = 2 * tx0 - 1;

    // ---- this thread's share of the halo tile: elements (pixel pix0 + u DP, channel quad cq), the same for every chunk ----
    constexpr int TOT = IH * IW * CQ;
    constexpr int PER = (TOT + NT - 1) / NT;
    constexpr int DP = NT / CQ;
    static_assert(NT % CQ == 0 && (PER - 1) * DP < IH * IW, "only a thread's LAST element can fall off the tile");
    const int cq = tid % CQ, pix0 = tid / CQ;
    unsigned voff[PER];                        // byte offset inside the sample (FB 2: of the pixel's 4-byte word), 2^31 when the element is padding
    unsigned voffg[FB == 1 ? PER : 1];         // FB 1 only: the pixel's word of d objective / d x_hat beside its channel quad
    unsigned bad = 0;                          // bit u: element u is padding (zero AFTER the activation)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pix = pix0 + u * DP, iy = pix / IW, ix = pix % IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        const bool ok = (u + 1 < PER || pix < IH * IW) && (unsigned)gy < (unsigned)AH && (unsigned)gx < (unsigned)AW;
        const unsigned gp = (unsigned)(gy * AW + gx);
        voff[u] = ok ? (fbb ? gp * 4u : (gp * (unsigned)CA + (unsigned)(cq * 4)) * 4u) : 0x80000000u;
        if (FB == 1) voffg[FB == 1 ? u : 0] = ok ? gp * 4u : 0x80000000u;
        bad |= (ok ? 0u : 1u) << u;
    }
    // LDS byte address of element u's hi store = lds_b + u DP LDH 2 (an instruction immediate), lo plane IH IW LDH 2 bytes further; a last element
    // that is off the tile lands in pixel 0's pad bytes (never read) instead of being branched around
    const unsigned lds_b = (unsigned)(pix0 * LDH + cq * 4) * 2u;
    const unsigned lds_last = (pix0 + (PER - 1) * DP < IH * IW) ? lds_b + (unsigned)((PER - 1) * DP * LDH) * 2u : (unsigned)CK * 2u;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<float*>(a.A) + (size_t)n * AH * AW * CA), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<float*>(a.xf.fb_dxhat) + (size_t)n * AH * AW), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<unsigned*>(a.xf.fb_bits) + (size_t)n * AH * AW), 0, 0x80000000u, 0x00020000);
    __syncthreads();

    // The activation tile of chunk ch+1 is fetched into registers while chunk ch is contracted (the tile's two dependent
    // HBM/L2 latencies were ~60 % of a workgroup's lifetime); conversion + LDS store happen once every wave has left chunk ch.
    uint4 pf[fbb ? 1 : PER];
    float pg[fb ? PER : 1];                  // fb mode: the pixel's d objective / d x_hat, fetched with the tile
    unsigned pb[fbb ? PER : 1];              // bits mode: the pixel's activation-pattern word
    auto issue_stage = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (fbb) pb[fbb ? u : 0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(brs, (int)voff[u], 0, 0);
            else pf[fbb ? 0 : u] = buf_load16(irs, voff[u], (unsigned)c0 * 4u);
            if (fb) pg[fb ? u : 0] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(grs, (int)(FB == 1 ? voffg[FB == 1 ? u : 0] : voff[u]), 0, 0));
        }
    };
    auto convert_stage = [&](int c0) __attribute__((always_inline)) {
        // this thread's channel quad is the same for every element (NT % CQ == 0): its table rows are read once per chunk, not once per element
        const float4 t_sc = (xf || fb) ? *reinterpret_cast<const float4*>(s_xf + c0 + cq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 t_sh = (xf || FB == 1) ? *reinterpret_cast<const float4*>(s_xf + XF_LDS_CH + c0 + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 t_wf = fb ? *reinterpret_cast<const float4*>(s_xf + 2 * XF_LDS_CH + c0 + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned bshift = (unsigned)(c0 + cq * 4);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint4 q = pf[fbb ? 0 : u];
            float4 t = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
            if (fbb) {
#pragma clang fp contract(off)      // (the products are rounded before the hi | lo split subtracts from them, as when a padding select stood between the two)
                // the same from the pattern word the fused forward epilogue left: the derivative side of every channel is one bit
                const float4 sc = t_sc, wf = t_wf;
                const float g = pg[fb ? u : 0];          // padding: 0 through the descriptor's range check -> the products are zeros
                const unsigned b = pb[fbb ? u : 0] >> bshift;
                t.x = g * wf.x * ((b & 1u) ? sc.x : sc.x * a.xf.alpha);
                t.y = g * wf.y * ((b & 2u) ? sc.y : sc.y * a.xf.alpha);
                t.z = g * wf.z * ((b & 4u) ? sc.z : sc.z * a.xf.alpha);
                t.w = g * wf.w * ((b & 8u) ? sc.w : sc.w * a.xf.alpha);
            } else if (fb) {
#pragma clang fp contract(off)
                // final-backward on load: t = dxhat[pixel] * wf * lrelu'(bn(c)) * scale  (see UadXform::fb_*)
                const float4 sc = t_sc, sh = t_sh, wf = t_wf;
                const float g = pg[fb ? u : 0];
                t.x = g * wf.x * (__builtin_fmaf(t.x, sc.x, sh.x) > 0.f ? sc.x : sc.x * a.xf.alpha);
                t.y = g * wf.y * (__builtin_fmaf(t.y, sc.y, sh.y) > 0.f ? sc.y : sc.y * a.xf.alpha);
                t.z = g * wf.z * (__builtin_fmaf(t.z, sc.z, sh.z) > 0.f ? sc.z : sc.z * a.xf.alpha);
                t.w = g * wf.w * (__builtin_fmaf(t.w, sc.w, sh.w) > 0.f ? sc.w : sc.w * a.xf.alpha);
            } else if (xf) {
                t = keep4(!((bad >> u) & 1u), xform4(t, t_sc, t_sh, a.xf.alpha));      // padding is zero AFTER the activation
            }
            uint2 pl[NPL];
            split_planes<NPL>(t, pl);
            const unsigned o = (u + 1 < PER) ? lds_b + (unsigned)(u * DP * LDH) * 2u : lds_last;
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(dsm + o + (unsigned)(p * PLANE) * 2u) = pl[p];
        }
    };
    issue_stage(ch0 * CK);
    for (int ch = ch0; ch < nchunks; ++ch) {
        const int c0 = ch * CK;
        if (ch > ch0) __syncthreads();
        convert_stage(c0);
        if (ch + 1 < nchunks) issue_stage(c0 + CK);
        __syncthreads();
        loadA(a0, 0);
        const bool more = ch + 1 < nchunks;
        auto unit = [&](const KFrag<NKS, NPL>& bc, KFrag<NKS, NPL>& bpf, const KFrag<NKS, NPL>& ac, KFrag<NKS, NPL>& an, const int t) {
            if (t + 3 < 25) loadB(bpf, t + 3, c0);
            else if (more) loadB(bpf, t + 3 - 25, c0 + CK);
            if (t + 1 < 25) loadA(an, t + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NKS; ++j) {
                acc0 = mfma_bf16(ac.p[0][j], bc.p[0][j], acc0);
                acc1 = mfma_bf16(ac.p[0][j], bc.p[1][j], acc1);
                acc2 = mfma_bf16(ac.p[1][j], bc.p[0][j], acc2);
                if constexpr (NPL == 3) {      // bf16x6: + a0 w2, a2 w0, a1 w1
                    acc1 = mfma_bf16(ac.p[0][j], bc.p[2][j], acc1);
                    acc2 = mfma_bf16(ac.p[2][j], bc.p[0][j], acc2);
                    acc1 = mfma_bf16(ac.p[1][j], bc.p[1][j], acc1);
                }
            }
        };
#pragma unroll
        for (int g4 = 0; g4 < 7; ++g4) {
            if (4 * g4 + 0 < 25) unit(b0, b3, a0, a1, 4 * g4 + 0);
            if (4 * g4 + 1 < 25) unit(b1, b0, a1, a0, 4 * g4 + 1);
            if (4 * g4 + 2 < 25) unit(b2, b1, a0, a1, 4 * g4 + 2);
            if (4 * g4 + 3 < 25) unit(b3, b2, a1, a0, 4 * g4 + 3);
        }
        // 25 = 6*4 + 1: the ring advanced by one slot; rotate so the next chunk starts at slot 0 again
        b0 = b1; b1 = b2; b2 = b3;
    }

    // ---- epilogue: transpose through a wave-private LDS tile, 16-byte accesses (see conv5_d16_kernel) ----
    __syncthreads();
    constexpr int EPI_LD = 36;
    static_assert((size_t)WGM * WGN * 32 * EPI_LD * 4 <= (size_t)NPL * IH * IW * LDH * 2, "epilogue tile fits in the activation tile");
    const bool bwd = (a.ep.kind == UAD_EPI_BWD_ACT);
    float* etile = s_epi + wave * 32 * EPI_LD;
    const int ec4 = (lane & 7) * 4, erow = lane >> 3;
    const int ecol = n0 + wn * 32 + ec4;
    v16f o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = acc0[r] + (acc1[r] + acc2[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) etile[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_LD + l31] = o[r];
    __builtin_amdgcn_wave_barrier();
    float4 v[4];
    size_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = erow + 8 * k;
        v[k] = *reinterpret_cast<const float4*>(etile + row * EPI_LD + ec4);
        const int mm = wm * 32 + row;
        off[k] = ((size_t)(n * d.HS + ty0 + mm / TW) * d.WS + tx0 + mm % TW) * Nn + ecol;
    }
    float* outp = a.Out;
    if (nsplit > 1) {
        if (!a.sk_counter) {
            float* slab = a.Out + (size_t)split * a.out_elems;
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(slab + off[k]) = v[k];
            return;
        }
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) sk_store16(srs, (unsigned)(((size_t)split * a.out_elems + off[k]) * 4), v[k]);
        const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) * (gridDim.z / nsplit) + blockIdx.z / nsplit;
        if (!sk_last_arriver(a.sk_counter + slot, nsplit, reinterpret_cast<int*>(s_red), tid)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < nsplit; sp += 2) {          // nsplit is a power of two >= 2; two slabs (8 loads) in flight
            float4 t[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) t[k][h] = sk_load16(srs, (unsigned)(((size_t)(sp + h) * a.out_elems + off[k]) * 4));
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) { v[k].x += t[k][h].x; v[k].y += t[k][h].y; v[k].z += t[k][h].z; v[k].w += t[k][h].w; }
        }
        outp = a.out_final;
        __syncthreads();      // s_red[0] carried the flag; it is reused below
    }
    if (!bwd) {
        float4 e_a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.ep.bias) e_a = *reinterpret_cast<const float4*>(a.ep.bias + ecol);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 t = v[k];
            t.x += e_a.x; t.y += e_a.y; t.z += e_a.z; t.w += e_a.w;
            if (a.ep.mul) { const float4 q = *reinterpret_cast<const float4*>(a.ep.mul + off[k]); t.x *= q.x; t.y *= q.y; t.z *= q.z; t.w *= q.w; }
            if (a.ep.add) { const float4 q = *reinterpret_cast<const float4*>(a.ep.add + off[k]); t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
            if (outp) *reinterpret_cast<float4*>(outp + off[k]) = t;
        }
        return;
    }
    float4 e_a = *reinterpret_cast<const float4*>(a.ep.escale + ecol);
    e_a.x *= a.ep.emult; e_a.y *= a.ep.emult; e_a.z *= a.ep.emult; e_a.w *= a.ep.emult;
    const float4 e_b = *reinterpret_cast<const float4*>(a.ep.eshift + ecol);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 cp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cp[k] = *reinterpret_cast<const float4*>(a.ep.cprev + off[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, cc[4] = {cp[k].x, cp[k].y, cp[k].z, cp[k].w};
        const float aa[4] = {e_a.x, e_a.y, e_a.z, e_a.w}, bb[4] = {e_b.x, e_b.y, e_b.z, e_b.w};
        float oo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bn = fmaf(aa[e], cc[e], bb[e]);
            const float dbn = bn > 0.f ? vv[e] : vv[e] * a.ep.ealpha;
            oo[e] = dbn * aa[e];
            s1[e] += dbn;
            s2[e] = fmaf(dbn, cc[e], s2[e]);
        }
        const float4 o4 = make_float4(oo[0], oo[1], oo[2], oo[3]);
        if (outp) *reinterpret_cast<float4*>(outp + off[k]) = o4;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t1 = s1[e], t2 = s2[e];
        t1 += __shfl_xor(t1, 8); t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
        t2 += __shfl_xor(t2, 8); t2 += __shfl_xor(t2, 16); t2 += __shfl_xor(t2, 32);
        if (lane < 8) {
            s_red[(wm * 2 + 0) * BN + wn * 32 + ec4 + e] = t1;
            s_red[(wm * 2 + 1) * BN + wn * 32 + ec4 + e] = t2;
        }
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid / BN, c = tid % BN;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) t += s_red[(w * 2 + which) * BN + c];
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (n0 + c < Nn) a.ep.colpart[(tile * 2 + which) * Nn + n0 + c] = t;
    }
}

// Launches without the AQL barrier bit (hipExtAnyOrderLaunch).  Packets of a queue are still DEQUEUED in order, so such a kernel starts once
// everything before its predecessor has completed (the predecessor's own barrier bit saw to that) -- but it does not wait for the predecessor
// itself.  The backward uses it for the two launches per layer that do not depend on the kernel enqueued right before them: a layer's data
// gradient behind its filter gradient (both read d loss / d c, neither reads the other's output), and the first filter gradient behind the
// forward's single-workgroup loss.finalize.  The successor's workgroups fill the CUs the predecessor's last workgroups are still draining:
// -0.7 % per step (profiles/r03_y_gpurun11/12.log); unlike a second stream (+36 %) the predecessor is dispatched in full first.
// UAD_NO_ANYORDER=1 (read in uad_model.hip) keeps every launch ordered.
// (g_any_order_next / g_any_order_w_next and the two launch macros: uad_gemm_common.h)
// The lane = pixel D-kind family (uad_conv16s.inc) is compiled in uad_gemm_d16s.hip; ConvGemmArgs is the same header-defined struct in both units and
// crosses the boundary as an untyped pointer (anonymous-namespace types are unit-local).
}  // namespace
bool uad_d16s_takes_v(const void* conv_gemm_args);
void uad_d16s_launch_v2(const void* conv_gemm_args, unsigned gx, unsigned gy, unsigned gz, int bn, int cst, bool any_order, hipStream_t st);      // uad_gemm_d16s.hip: two planes
void uad_d16s_launch_v3(const void* conv_gemm_args, unsigned gx, unsigned gy, unsigned gz, int bn, int cst, bool any_order, hipStream_t st);      // uad_gemm_d16s3.hip: three
namespace {
inline bool conv5_d16s_takes(const ConvGemmArgs& a) { return uad_d16s_takes_v(&a); }
inline void launch_conv5_d16s_any(const ConvGemmArgs& a, dim3 grid, int bn, int cst, hipStream_t st) {
    const bool any = g_any_order_next;
    g_any_order_next = false;
    if (a.npl == 3) uad_d16s_launch_v3(&a, grid.x, grid.y, grid.z, bn, cst, any, st); else uad_d16s_launch_v2(&a, grid.x, grid.y, grid.z, bn, cst, any, st);
}

template <int TH, int TW, int CK, int WGM, int WGN, int NPL = 2>
constexpr size_t conv5_f16_lds_bytes() {
    return (size_t)NPL * (2 * TH + 3) * (2 * TW + 3) * (CK + 8) * 2 + (size_t)3 * XF_LDS_CH * 4 + (size_t)WGM * 2 * 32 * WGN * 4;
}

template <int TH, int TW, int CK, int WGM, int WGN, int FB, bool XF, int NPL>
void launch_conv5_f16_v(const ConvGemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = conv5_f16_lds_bytes<TH, TW, CK, WGM, WGN, NPL>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_f16_kernel<TH, TW, CK, WGM, WGN, FB, XF, NPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    UAD_SPATIAL_LAUNCH((conv5_f16_kernel<TH, TW, CK, WGM, WGN, FB, XF, NPL>), grid, dim3(64 * WGM * WGN), lds, st, a);
}
template <int TH, int TW, int CK, int WGM, int WGN, int NPL = 2>
void launch_conv5_f16(const ConvGemmArgs& a, dim3 grid, hipStream_t st) {
    if (a.xf.fb_bits || a.xf.fb_dxhat) {
        if (!a.xf.scale) { fprintf(stderr, "uad: final-backward on load needs the block's scale / shift tables\n"); abort(); }
        if (a.xf.fb_bits && a.CA > 32) { fprintf(stderr, "uad: the pattern-word form holds one bit per channel of a 32-bit word (CA = %d)\n", a.CA); abort(); }
        if (a.xf.fb_bits) launch_conv5_f16_v<TH, TW, CK, WGM, WGN, 2, true, NPL>(a, grid, st);
        else launch_conv5_f16_v<TH, TW, CK, WGM, WGN, 1, true, NPL>(a, grid, st);
    } else if (a.xf.scale) launch_conv5_f16_v<TH, TW, CK, WGM, WGN, 0, true, NPL>(a, grid, st);
    else launch_conv5_f16_v<TH, TW, CK, WGM, WGN, 0, false, NPL>(a, grid, st);
}

template <int TH, int TW, int CST, int WGM, int WGN>
constexpr size_t conv5_d16_lds_bytes() {
    return (size_t)2 * (TH + 2) * (TW + 2) * (CST + 8) * 2 + (size_t)2 * XF_LDS_CH * 4 + (size_t)WGM * 2 * 32 * WGN * 4 +
           (size_t)WGM * WGN * 32 * 36 * 4 + (size_t)5 * 4 * TH * TW * 4;
}
template <int TH, int TW, int CST, int WGM, int WGN>
void launch_conv5_d16(const ConvGemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = conv5_d16_lds_bytes<TH, TW, CST, WGM, WGN>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_d16_kernel<TH, TW, CST, WGM, WGN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv5_d16_kernel<TH, TW, CST, WGM, WGN>), grid, dim3(64 * WGM * WGN), lds, st, a);
}

template <int TH, int TW, int CK, int WGM, int WGN, int KIND>
constexpr size_t conv5_bf16_lds_bytes() {
    return (size_t)2 * ((KIND == KIND_F) ? (2 * TH + 3) * (2 * TW + 3) : (TH + 2) * (TW + 2)) * (CK + 8) * 2 +
           (size_t)2 * XF_LDS_CH * 4 + (size_t)WGM * 2 * 32 * WGN * 4;
}

template <int TH, int TW, int CK, int WGM, int WGN, int KIND>
void launch_conv5_bf16(const ConvGemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = conv5_bf16_lds_bytes<TH, TW, CK, WGM, WGN, KIND>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_bf16_kernel<TH, TW, CK, WGM, WGN, KIND>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv5_bf16_kernel<TH, TW, CK, WGM, WGN, KIND>), grid, dim3(64 * WGM * WGN), lds, st, a);
}

// bf16 hi|lo planes for the bf16x3 kernels: Wf16[plane][tap][cb/8][cs][8], Wd16[plane][tap][cs/8][cb][8] (ushort),
// each tensor at 2*off with its lo plane `count` elements behind the hi plane
__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// Sums S split-K slabs (fixed order -> deterministic) and applies the epilogue.  Block = 64 rows x 64 columns:
// 16 column-quad lanes x 16 row lanes, 4 rows per thread.
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const float* __restrict__ slabs, int S, long long out_elems,
                                                              int rows, int Nn, UadEpilogue ep, float* __restrict__ out) {
    __shared__ float red[2][16][64];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.y * 64 + cq * 4;
    const bool colok = col < Nn;
    const bool bwd = ep.kind == UAD_EPI_BWD_ACT;
    float4 ca = make_float4(0, 0, 0, 0), cb = ca;
    if (colok) {
        if (!bwd) { if (ep.bias) ca = *reinterpret_cast<const float4*>(ep.bias + col); }
        else {
            ca = *reinterpret_cast<const float4*>(ep.escale + col);
            ca.x *= ep.emult; ca.y *= ep.emult; ca.z *= ep.emult; ca.w *= ep.emult;
            cb = *reinterpret_cast<const float4*>(ep.eshift + col);
        }
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = blockIdx.x * 64 + u * 16 + rl;
        if (row >= rows || !colok) continue;
        const size_t off = (size_t)row * Nn + col;
        float4 v = *reinterpret_cast<const float4*>(slabs + off);
        for (int s = 1; s < S; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(slabs + (size_t)s * out_elems + off);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (!bwd) {
            v.x += ca.x; v.y += ca.y; v.z += ca.z; v.w += ca.w;
            if (ep.mul) { const float4 t = *reinterpret_cast<const float4*>(ep.mul + off); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
            if (ep.add) { const float4 t = *reinterpret_cast<const float4*>(ep.add + off); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            *reinterpret_cast<float4*>(out + off) = v;
        } else {
            const float4 c = *reinterpret_cast<const float4*>(ep.cprev + off);
            const float vv[4] = {v.x, v.y, v.z, v.w}, cc[4] = {c.x, c.y, c.z, c.w};
            const float aa[4] = {ca.x, ca.y, ca.z, ca.w}, bb[4] = {cb.x, cb.y, cb.z, cb.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bn = fmaf(aa[e], cc[e], bb[e]);
                const float dbn = bn > 0.f ? vv[e] : vv[e] * ep.ealpha;
                o[e] = dbn * aa[e];
                s1[e] += dbn;
                s2[e] = fmaf(dbn, cc[e], s2[e]);
            }
            *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (bwd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][rl][cq * 4 + e] = s1[e]; red[1][rl][cq * 4 + e] = s2[e]; }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[which][k][c];
            const int gc = blockIdx.y * 64 + c;
            if (gc < Nn) ep.colpart[((size_t)blockIdx.x * 2 + which) * Nn + gc] = t;
        }
    }
}

// Weight re-layout for the spatial kernels.  One thread per source element W[tap][cb][cs] of any of up to 8 tensors:
//   F-pack: Wp[tap][cb/4][cs][cb%4]     D-pack: Wq[tap][cs/4][cb][cs%4]      (both at the tensor's own flat offset)
struct PackDesc { long long off[16]; int cb[16], cs[16], count[16]; int n; };
__global__ void pack_weights_kernel(const float* __restrict__ W, float* __restrict__ Wf, float* __restrict__ Wd, PackDesc pd) {
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int t = 0;
    while (t < pd.n && gid >= pd.count[t]) { gid -= pd.count[t]; ++t; }
    if (t >= pd.n) return;
    const int CB = pd.cb[t], CS = pd.cs[t];
    const int cs = (int)(gid % CS);
    const int r = (int)(gid / CS);
    const int cb = r % CB, tap = r / CB;
    const float v = W[pd.off[t] + gid];
    Wf[pd.off[t] + ((size_t)(tap * (CB / 4) + cb / 4) * CS + cs) * 4 + (cb & 3)] = v;
    Wd[pd.off[t] + ((size_t)(tap * (CS / 4) + cs / 4) * CB + cb) * 4 + (cs & 3)] = v;
}

__global__ void pack_weights_bf16_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wf, unsigned short* __restrict__ Wd, PackDesc pd) {
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int t = 0;
    while (t < pd.n && gid >= pd.count[t]) { gid -= pd.count[t]; ++t; }
    if (t >= pd.n) return;
    const int CB = pd.cb[t], CS = pd.cs[t];
    const int cs = (int)(gid % CS);
    const int r = (int)(gid / CS);
    const int cb = r % CB, tap = r / CB;
    const float v = W[pd.off[t] + gid];
    const unsigned hi = bf16_rne_bits(v);
    const unsigned lo = bf16_rne_bits(v - __uint_as_float(hi << 16));
    const size_t base = 2 * (size_t)pd.off[t], cnt = (size_t)pd.count[t];
    const size_t fi = ((size_t)(tap * (CB / 8) + cb / 8) * CS + cs) * 8 + (cb & 7);
    const size_t di = ((size_t)(tap * (CS / 8) + cs / 8) * CB + cb) * 8 + (cs & 7);
    Wf[base + fi] = (unsigned short)hi; Wf[base + cnt + fi] = (unsigned short)lo;
    Wd[base + di] = (unsigned short)hi; Wd[base + cnt + di] = (unsigned short)lo;
}

// Three planes (bf16x6, uad_convk16.inc): x = h + m + l; planes of a tensor at ushort index 4 * off + p * count in buffers of 4 * nparams ushorts
// (4, not 3: a tensor's first plane then starts 8 * off bytes in -- dword-aligned for the fragment loads whatever the offset's parity).
__global__ void pack_weights_bf16_3p_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wf, unsigned short* __restrict__ Wd, PackDesc pd) {
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int t = 0;
    while (t < pd.n && gid >= pd.count[t]) { gid -= pd.count[t]; ++t; }
    if (t >= pd.n) return;
    const int CB = pd.cb[t], CS = pd.cs[t];
    const int cs = (int)(gid % CS);
    const int r = (int)(gid / CS);
    const int cb = r % CB, tap = r / CB;
    const float v = W[pd.off[t] + gid];
    const unsigned h = bf16_rne_bits(v);
    const float r1 = v - __uint_as_float(h << 16);
    const unsigned m = bf16_rne_bits(r1);
    const unsigned l = bf16_rne_bits(r1 - __uint_as_float(m << 16));
    const size_t base = 4 * (size_t)pd.off[t], cnt = (size_t)pd.count[t];
    const size_t fi = ((size_t)(tap * (CB / 8) + cb / 8) * CS + cs) * 8 + (cb & 7);
    const size_t di = ((size_t)(tap * (CS / 8) + cs / 8) * CB + cb) * 8 + (cs & 7);
    Wf[base + fi] = (unsigned short)h; Wf[base + cnt + fi] = (unsigned short)m; Wf[base + 2 * cnt + fi] = (unsigned short)l;
    Wd[base + di] = (unsigned short)h; Wd[base + cnt + di] = (unsigned short)m; Wd[base + 2 * cnt + di] = (unsigned short)l;
}

// eligibility + tile choice of the spatial kernels (shared by the launchers and the *_tiles() queries)
// The same packing, one 16-byte group of a packed layout per thread (blockIdx.y: 0 = the F layout, 8 consecutive cb of one (tap, cs); 1 = the D
// layout, 8 consecutive cs of one (tap, cb)): coalesced 16-byte stores instead of four scattered 2-byte stores per element (the repack runs on the
// side stream behind the optimizer step; at 20 us it outlasted the main-stream work before the first packed-weight consumer).  Same bits.
// A tensor's offset in the flat parameter vector need not be a multiple of four floats (the ResNet graph's critic starts one float after the generator's
// one-channel 1x1 convolution: every critic tensor sat at offset 1 mod 4 and took the per-element kernels -- 133 + 81 us per phase in the round-6 trace):
// the 16-byte accesses are typed dword-aligned, which the hardware serves at any dword address.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NPL>      // 2: hi | lo planes at ushort index 2 * off (bf16x3); 3: h | m | l at 4 * off (bf16x6; round 6 -- the per-element 3p kernel was 4.5 % of a configs[3] iteration)
__global__ void __launch_bounds__(256) pack_weights_bf16_v8_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wf, unsigned short* __restrict__ Wd, PackDesc pd) {
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // group index inside the tensor list
    int t = 0;
    while (t < pd.n && g >= pd.count[t] / 8) { g -= pd.count[t] / 8; ++t; }
    if (t >= pd.n) return;
    const int CB = pd.cb[t], CS = pd.cs[t];
    const float* w = W + pd.off[t];
    float v[8];
    size_t dst;      // first element of the group inside the tensor's packed plane
    if (blockIdx.y == 0) {
        const int cs = (int)(g % CS);
        const int r = (int)(g / CS);
        const int cb8 = r % (CB / 8), tap = r / (CB / 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[((size_t)tap * CB + cb8 * 8 + e) * CS + cs];
        dst = ((size_t)(tap * (CB / 8) + cb8) * CS + cs) * 8;
    } else {
        const int cb = (int)(g % CB);
        const int r = (int)(g / CB);
        const int cs8 = r % (CS / 8), tap = r / (CS / 8);
        const f4u a = *reinterpret_cast<const f4u*>(w + ((size_t)tap * CB + cb) * CS + cs8 * 8);
        const f4u b = *reinterpret_cast<const f4u*>(w + ((size_t)tap * CB + cb) * CS + cs8 * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        dst = ((size_t)(tap * (CS / 8) + cs8) * CB + cb) * 8;
    }
    unsigned pl[NPL][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = v[e];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {      // same arithmetic as the per-element kernels: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
            pl[q][e] = bf16_rne_bits(r);
            r -= __uint_as_float(pl[q][e] << 16);
        }
    }
    unsigned short* out = blockIdx.y == 0 ? Wf : Wd;
    const size_t base = (NPL == 3 ? 4 : 2) * (size_t)pd.off[t], cnt = (size_t)pd.count[t];
#pragma unroll
    for (int q = 0; q < NPL; ++q)
        *reinterpret_cast<u4u*>(out + base + q * cnt + dst) = u4u{pl[q][0] | (pl[q][1] << 16), pl[q][2] | (pl[q][3] << 16), pl[q][4] | (pl[q][5] << 16), pl[q][6] | (pl[q][7] << 16)};
}

struct SpatialChoice { bool ok; int TH, TW, BN, CK; };
inline SpatialChoice choose_spatial(const UadConvDesc& d, int CA, int Nn, bool f_type = true) {
    SpatialChoice c{false, 0, 0, 0, 0};
    if (!(d.KS == 5 && d.S == 2 && d.P == 1)) return c;
    if (d.HB != 2 * d.HS || d.WB != 2 * d.WS) return c;
    if (CA > XF_LDS_CH) return c;
    if (Nn % 64 == 0 && CA % 32 == 0 && d.HS % 8 == 0 && d.WS % 8 == 0) { c = SpatialChoice{true, 8, 8, 64, 32}; return c; }
    // 8x16 tile, 32 output channels: the D-type halo (10x18) is small enough for 32-channel chunks, the F-type one is not
    if (Nn % 32 == 0 && CA % 16 == 0 && d.HS % 8 == 0 && d.WS % 16 == 0) {
        c = SpatialChoice{true, 8, 16, 32, (!f_type && CA % 32 == 0) ? 32 : 16};
        return c;
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// W-type: filter gradient.  GEMM M = (tap, cb) flattened, N = cs, K = positions (split over blockIdx.z).
// ------------------------------------------------------------------------------------------------
struct ConvWArgs {
    const float* big;
    const float* small_;
    float* partial;
    UadXform xfb, xfs;
    UadConvDesc d;
    int Mtot;  // KS*KS*CB
    int Kt;    // N*HS*WS
    int kper;  // positions per split (multiple of BK)
    int lws, lhs;
    unsigned long long* dbgbuf;   // UAD_DBG & 32: per-workgroup start / end clocks
    int npl = 2;   // bf16 planes per operand of the k5 s2 kernel (2: bf16x3 products, 3: bf16x6)
    int abl = 0;   // UAD_W_ABL, kernel-tuning ablations (results are wrong): 1 no big-tile LDS stores | 2 no MFMAs | 4 no big-tile global loads | 8 no slab stores;
                   // 32 (results stay right): every workgroup sleeps before its slab store (ordering stress test)
};

template <int BM, int BN, int BK, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_w_kernel(const ConvWArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_EL = BK * LDA, B_EL = BK * LDB, STAGE = A_EL + B_EL;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const UadConvDesc& d = a.d;
    const int kbeg = blockIdx.z * a.kper;
    const int kend = min(kbeg + a.kper, a.Kt);
    const int nk = (kend - kbeg + BK - 1) / BK;

    constexpr int TPRA = BM / 4, RPA = NT / TPRA, PA = (BK + RPA - 1) / RPA;
    const int acol = (tid % TPRA) * 4, arow0 = tid / TPRA;
    const int mcol = m0 + acol;
    const bool acolok = mcol < a.Mtot;
    const int tapA = acolok ? mcol / d.CB : 0;
    const int cbA = acolok ? mcol - tapA * d.CB : 0;
    const int kyA = tapA / d.KS, kxA = tapA - kyA * d.KS;
    float4 scA = make_float4(1, 1, 1, 1), shA = make_float4(0, 0, 0, 0);
    const bool xfa = a.xfb.scale != nullptr;
    if (xfa && acolok) {
        scA = *reinterpret_cast<const float4*>(a.xfb.scale + cbA);
        shA = *reinterpret_cast<const float4*>(a.xfb.shift + cbA);
        scA.x *= a.xfb.mult; scA.y *= a.xfb.mult; scA.z *= a.xfb.mult; scA.w *= a.xfb.mult;
    }
    constexpr int TPRB = BN / 4, RPB = NT / TPRB, PB = (BK + RPB - 1) / RPB;
    const int bcol = (tid % TPRB) * 4, brow0 = tid / TPRB;
    const int ncol = n0 + bcol;
    const bool bcolok = ncol < d.CS;
    float4 scB = make_float4(1, 1, 1, 1), shB = make_float4(0, 0, 0, 0);
    const bool xfs = a.xfs.scale != nullptr;
    if (xfs && bcolok) {
        scB = *reinterpret_cast<const float4*>(a.xfs.scale + ncol);
        shB = *reinterpret_cast<const float4*>(a.xfs.shift + ncol);
        scB.x *= a.xfs.mult; scB.y *= a.xfs.mult; scB.z *= a.xfs.mult; scB.w *= a.xfs.mult;
    }

    float4 va[PA], vb[PB];
    unsigned okA = 0, okB = 0;
    auto load_tiles = [&](int kpos) {
        okA = 0; okB = 0;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = arow0 + q * RPA;
            const int pos = kpos + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < BK && pos < kend && acolok) {
                int n, i, j;
                decode_pos(pos, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                const int y = d.S * i - d.P + kyA, x = d.S * j - d.P + kxA;
                if ((unsigned)y < (unsigned)d.HB && (unsigned)x < (unsigned)d.WB) {
                    v = *reinterpret_cast<const float4*>(a.big + ((size_t)(n * d.HB + y) * d.WB + x) * d.CB + cbA);
                    okA |= 1u << q;
                }
            }
            va[q] = v;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int r = brow0 + q * RPB;
            const int pos = kpos + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < BK && pos < kend && bcolok) {
                v = *reinterpret_cast<const float4*>(a.small_ + (size_t)pos * d.CS + ncol);
                okB |= 1u << q;
            }
            vb[q] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + A_EL;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = arow0 + q * RPA;
            float4 v = va[q];
            if (xfa && ((okA >> q) & 1u)) v = xform4(v, scA, shA, a.xfb.alpha);
            if (r < BK) *reinterpret_cast<float4*>(sA + r * LDA + acol) = v;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int r = brow0 + q * RPB;
            float4 v = vb[q];
            if (xfs && ((okB >> q) & 1u)) v = xform4(v, scB, shB, a.xfs.alpha);
            if (r < BK) *reinterpret_cast<float4*>(sB + r * LDB + bcol) = v;
        }
    };

    v16f acc[FM][FN];
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[im][jn][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) load_tiles(kbeg + (ks + 1) * BK);
        const float* sA = smem + buf * STAGE;
        const float* sB = sA + A_EL;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = kk * 8 + 4 * lh + s;
                float av[FM], bv[FN];
#pragma unroll
                for (int im = 0; im < FM; ++im) av[im] = sA[k * LDA + wm * WTM + im * 32 + l31];
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) bv[jn] = sB[k * LDB + wn * WTN + jn * 32 + l31];
#pragma unroll
                for (int im = 0; im < FM; ++im)
#pragma unroll
                    for (int jn = 0; jn < FN; ++jn)
                        acc[im][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[im], bv[jn], acc[im][jn], 0, 0, 0);
            }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float* out = a.partial + (size_t)blockIdx.z * a.Mtot * d.CS;
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= a.Mtot) continue;
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                const int col = n0 + wn * WTN + jn * 32 + l31;
                if (col < d.CS) out[(size_t)m * d.CS + col] = acc[im][jn][r];
            }
        }
}


// ---- generic filter gradient in bf16x3 math (UAD_MATH_BF16X3_ALL): both operands arrive position-major ([k = position][channels]) and the
// matrix cores want channel-major rows with K contiguous, so each thread owns PAIRS of adjacent positions and stores (k, k+1) as one
// 32-bit word per channel into the hi | lo bf16 planes; a K = 16 slice is three v_mfma_f32_32x32x16_bf16.  Identity activation-on-load only.
template <int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_w16_kernel(const ConvWArgs a) {
    constexpr int BK = 32, NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
    constexpr int LDK = BK + 8;
    constexpr int A_EL = BM * LDK, B_EL = BN * LDK, STAGE = 2 * A_EL + 2 * B_EL;     // A hi | A lo | B hi | B lo
    __shared__ __attribute__((aligned(16))) unsigned short smem16[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const UadConvDesc& d = a.d;
    const int kbeg = blockIdx.z * a.kper;
    const int kend = min(kbeg + a.kper, a.Kt);
    const int nk = (kend - kbeg + BK - 1) / BK;

    constexpr int TPRA = BM / 4, RPA = NT / TPRA, PA = BK / RPA;
    constexpr int TPRB = BN / 4, RPB = NT / TPRB, PB = BK / RPB;
    static_assert(PA % 2 == 0 && PB % 2 == 0 && PA * RPA == BK && PB * RPB == BK, "each thread owns pairs of adjacent positions");
    const int acol = (tid % TPRA) * 4, arow0 = tid / TPRA;
    const int bcol = (tid % TPRB) * 4, brow0 = tid / TPRB;
    auto arow = [&](int q) { return 2 * arow0 + (q & 1) + 2 * RPA * (q >> 1); };
    auto brow = [&](int q) { return 2 * brow0 + (q & 1) + 2 * RPB * (q >> 1); };
    const int mcol = m0 + acol;
    const bool acolok = mcol < a.Mtot;
    const int tapA = acolok ? mcol / d.CB : 0;
    const int cbA = acolok ? mcol - tapA * d.CB : 0;
    const int kyA = tapA / d.KS, kxA = tapA - kyA * d.KS;
    const int ncol = n0 + bcol;
    const bool bcolok = ncol < d.CS;

    float4 va[PA], vb[PB];
    auto load_tiles = [&](int kpos) {
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int pos = kpos + arow(q);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < kend && acolok) {
                int n, i, j;
                decode_pos(pos, d.HS, d.WS, a.lhs, a.lws, n, i, j);
                const int y = d.S * i - d.P + kyA, x = d.S * j - d.P + kxA;
                if ((unsigned)y < (unsigned)d.HB && (unsigned)x < (unsigned)d.WB)
                    v = *reinterpret_cast<const float4*>(a.big + ((size_t)(n * d.HB + y) * d.WB + x) * d.CB + cbA);
            }
            va[q] = v;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int pos = kpos + brow(q);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < kend && bcolok) v = *reinterpret_cast<const float4*>(a.small_ + (size_t)pos * d.CS + ncol);
            vb[q] = v;
        }
    };
    // (k, k+1) of four channels -> four 32-bit words per plane
    auto store_pair = [&](unsigned short* ph, unsigned short* pl, int col, int k, const float4 v0, const float4 v1) {
        uint2 h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        const unsigned hh0[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
        const unsigned hh1[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
        const unsigned ll0[4] = {l0.x & 0xFFFFu, l0.x >> 16, l0.y & 0xFFFFu, l0.y >> 16};
        const unsigned ll1[4] = {l1.x & 0xFFFFu, l1.x >> 16, l1.y & 0xFFFFu, l1.y >> 16};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<unsigned*>(ph + (col + i) * LDK + k) = hh0[i] | (hh1[i] << 16);
            *reinterpret_cast<unsigned*>(pl + (col + i) * LDK + k) = ll0[i] | (ll1[i] << 16);
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned short* sAh = smem16 + buf * STAGE;
        unsigned short* sAl = sAh + A_EL;
        unsigned short* sBh = sAl + A_EL;
        unsigned short* sBl = sBh + B_EL;
#pragma unroll
        for (int q = 0; q < PA; q += 2) store_pair(sAh, sAl, acol, arow(q), va[q], va[q + 1]);
#pragma unroll
        for (int q = 0; q < PB; q += 2) store_pair(sBh, sBl, bcol, brow(q), vb[q], vb[q + 1]);
    };

    v16f acc[FM][FN];
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int jn = 0; jn < FN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[im][jn][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) load_tiles(kbeg + (ks + 1) * BK);
        const unsigned short* sAh = smem16 + buf * STAGE;
        const unsigned short* sAl = sAh + A_EL;
        const unsigned short* sBh = sAl + A_EL;
        const unsigned short* sBl = sBh + B_EL;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            uint4 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
            for (int im = 0; im < FM; ++im) {
                const int o = (wm * WTM + im * 32 + l31) * LDK + kk * 16 + 8 * lh;
                ah[im] = *reinterpret_cast<const uint4*>(sAh + o);
                al[im] = *reinterpret_cast<const uint4*>(sAl + o);
            }
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                const int o = (wn * WTN + jn * 32 + l31) * LDK + kk * 16 + 8 * lh;
                bh[jn] = *reinterpret_cast<const uint4*>(sBh + o);
                bl[jn] = *reinterpret_cast<const uint4*>(sBl + o);
            }
#pragma unroll
            for (int im = 0; im < FM; ++im)
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) {
                    acc[im][jn] = mfma_bf16(ah[im], bh[jn], acc[im][jn]);
                    acc[im][jn] = mfma_bf16(ah[im], bl[jn], acc[im][jn]);
                    acc[im][jn] = mfma_bf16(al[im], bh[jn], acc[im][jn]);
                }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float* out = a.partial + (size_t)blockIdx.z * a.Mtot * d.CS;
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= a.Mtot) continue;
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                const int col = n0 + wn * WTN + jn * 32 + l31;
                if (col < d.CS) out[(size_t)m * d.CS + col] = acc[im][jn][r];
            }
        }
}

#include "uad_conv5_f32w.inc"      // conv5_w_kernel: the exact-fp32 k5 s2 filter gradient

// Transpose-read form (round 4).  gfx950's ds_read_b64_tr_b16 gathers, per 16-lane group, four rows of 16 bf16 columns -- each lane supplies the
// 8-byte-aligned address of four CONTIGUOUS columns of one row, row stride free -- and hands lane j column j's four row values (probe:
// tools/tr_probe.hip, profiles/r04_g_tr_probe.log).  With rows = positions and columns = channels that IS the W-kind MFMA fragment (lane = channel,
// eight consecutive positions per lane = two reads), read straight from a PIXEL-major tile: lane (i = l & 15, chalf = (l >> 4) & 1, lh = l >> 5) points
// at pixel (2 (2 js + lh) + ky, 2 (4 t + i / 4) + kx), channels 16 chalf + 4 (i & 3) .. + 3.  The big halo and the small tile are therefore staged
// exactly as the F-kind kernels stage theirs -- two 8-byte LDS stores per float4 instead of eight 2-byte scatter stores with their shifts -- and the
// taps need no segment shifting (v_alignbit) at all.  Bank check: the four rows of a group sit 160 B (big, pixel stride 2) / 144 B (small) apart,
// 32 B each: dword banks [0-7] [40-47] [16-23] [56-63] / [0-7] [36-43] [8-15] [44-51], disjoint.  Tap-to-wave assignment: wave w of a cs block owns kernel row w's
// five taps + tap (4, w) + a quarter of tap (4, 4) (folded through LDS in a fixed order); one [25][CB][CS] slab per workgroup, summed in split order by
// reduce_partials_kernel.  (Its predecessors -- the pixel-major gather kernel of round 2 and the channel-major scatter kernel of round 3, same bits -- were
// dropped in round 6; git history has them.)
template <int NCSB, bool FBB, bool XFA, bool XFS, int NPL = 2, bool DB = false>      // NPL: bf16 planes per operand (2: bf16x3 products, 3: bf16x6); DB: double-buffered tiles
__global__ void __launch_bounds__(256 * NCSB) __attribute__((amdgpu_waves_per_eu(2)))
conv5_w_bf16_tr_kernel(const ConvWArgs a, int tiles_per_split, int total_tiles) {
    // Round 6 ("VALU diet"): the staging half of this kernel issued ~45 VALU instructions per 16 bytes staged, a third of them 64-bit pointer
    // arithmetic and border compares redone per element in BOTH the load-issue and the convert phase, chopped into ~40 basic blocks per tile by
    // run-time switches (activation-on-load or not, tuning ablations, phase clocks).  Now: the switches are template parameters (one straight-line
    // block per phase); every tile-invariant quantity of a thread's elements is computed ONCE before the tile loop -- per-element byte offsets
    // (the loads go through buffer descriptors: wave-uniform base + wave-uniform tile offset + 32-bit lane offset, no 64-bit VALU), four
    // border-class bit masks, the transform rows of the thread's channel quad (registers, not LDS re-reads), LDS store addresses (one base +
    // instruction immediates); the tile coordinates advance incrementally (no divisions); the halo's zero padding comes from the descriptor's
    // range check (an out-of-image element's offset is replaced by 2^31 = the descriptor's size) and costs a select per VALUE only where an activation is applied on load.
    // Same elements to the same threads, same arithmetic per element, same MFMA order: the slabs are bit-identical to round 5's.
    constexpr int TH = 8, TW = 8, IH = 2 * TH + 3, IW = 2 * TW + 3, CK = 32, CQ = CK / 4, NT = 256 * NCSB, CSQ = 8 * NCSB;
    constexpr int LDH = CK + 8;                   // ushorts per big pixel (80 B: 64 B of channels + 16 B pad)
    constexpr int BIGP = IH * IW * LDH;           // ushorts per plane
    constexpr int LDQ = 32 * NCSB + 8;            // ushorts per small position (144 B at NCSB = 2)
    static_assert(!(FBB && XFA), "the pattern-word form replaces the big operand's transform");
    typedef short v4s __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    // DB (round 6, the eight-wave 64-column instances in bf16x3): both tiles double-buffered -- tile t + 1 is converted into the other buffer between the
    // second and the third position block of tile t's MFMAs, the global loads of tile t + 2 follow it, ONE barrier per tile.  (Round 4 measured this neutral on
    // the pre-diet kernel, profiles/r04_h_w_tr_ab.log; on the straight-line tile body it is what it was worth on the k3 kernel: profiles/r06_g_k3_forms.md.)
    // Same elements, same products, same order: bit-identical slabs.
    constexpr int SMALLP = TH * TW * LDQ;         // ushorts per small plane
    constexpr int BUFU = NPL * BIGP + NPL * SMALLP;      // ushorts per buffer
    unsigned short* bPl = reinterpret_cast<unsigned short*>(dsm);       // NPL big planes, then NPL small planes (DB: twice)
    unsigned short* sPlT = bPl + NPL * BIGP;
    float* sRed = reinterpret_cast<float*>(dsm);   // reused after the tile loop

    const unsigned long long dbg_t0 = a.dbgbuf ? wall_clock64() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int wave = wave_all & 3;   // kernel row of this wave's taps
    const int csb = wave_all >> 2;    // its cs block
    const int l31 = lane & 31, lh = lane >> 5;
    const UadConvDesc& d = a.d;
    const int cb0 = blockIdx.x * 32, cs0 = blockIdx.y * 32 * NCSB;
    const int tilesx = d.WS / TW, tilesy = d.HS / TH;
    const int t_begin = blockIdx.z * tiles_per_split;
    const int t_end = min(t_begin + tiles_per_split, total_tiles);

    // ---- this thread's share of a tile: elements (pixel pix0 + u DP, channel quad cq), u < PER, of the 19 x 19 halo; (position pos0 + 32 u,
    // channel quad csq), u < SPER, of the 8 x 8 small tile.  Everything below is the same for every tile. ----
    constexpr int TOT = IH * IW * CQ;
    constexpr int PER = (TOT + NT - 1) / NT;          // 16-byte elements of the big tile per thread: 12 (NT 256) or 6 (NT 512)
    constexpr int DP = NT / CQ;                       // pixels between a thread's consecutive elements
    constexpr int SPER = TH * TW * CSQ / NT;          // 16-byte elements of the small tile per thread (2)
    static_assert(NT % CQ == 0 && NT % CSQ == 0 && (PER - 1) * DP < IH * IW, "only a thread's LAST element can fall off the tile");
    const int cq = tid % CQ, pix0 = tid / CQ;
    const int csq = tid % CSQ, pos0 = tid / CSQ;
    unsigned voff[PER];                               // byte offset of element u from the halo's origin pixel (FBB: of its pixel's 4-byte word)
    unsigned clsT = 0, clsB = 0, clsL = 0, clsR = 0;  // bit u: element u lies in halo row 0 | rows 17, 18 | column 0 | columns 17, 18
    const unsigned pixbytes = FBB ? 4u : (unsigned)d.CB * 4u;
    const unsigned chanbytes = FBB ? 0u : (unsigned)(cb0 + cq * 4) * 4u;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int p = pix0 + u * DP, iy = p / IW, ix = p % IW;
        const bool valid = u + 1 < PER || p < IH * IW;
        voff[u] = valid ? (unsigned)(iy * d.WB + ix) * pixbytes + chanbytes : 0x80000000u;
        clsT |= (iy == 0 ? 1u : 0u) << u; clsB |= (iy >= IH - 2 ? 1u : 0u) << u;
        clsL |= (ix == 0 ? 1u : 0u) << u; clsR |= (ix >= IW - 2 ? 1u : 0u) << u;
    }
    // LDS byte address of element u's hi store = lds_b + u * DP * LDH * 2 (an instruction immediate); the lo plane is BIGP * 2 bytes further.  A thread
    // whose last element is off the tile stores it into pixel 0's pad bytes (never read) instead of branching around the store.
    const unsigned lds_b = (unsigned)(pix0 * LDH + cq * 4) * 2u;
    const bool last_valid = pix0 + (PER - 1) * DP < IH * IW;
    const unsigned lds_last = last_valid ? lds_b + (unsigned)((PER - 1) * DP * LDH) * 2u : (unsigned)CK * 2u;
    unsigned svoff[SPER];
#pragma unroll
    for (int u = 0; u < SPER; ++u) {
        const int pos = pos0 + u * (NT / CSQ);
        svoff[u] = (unsigned)(((pos / TW) * d.WS + (pos % TW)) * d.CS + cs0 + csq * 4) * 4u;
    }
    const unsigned slds_b = (unsigned)(pos0 * LDQ + csq * 4) * 2u;       // + u * 32 * LDQ * 2; lo plane TH * TW * LDQ * 2 bytes further

    // activation-on-load rows of this thread's channel quads (registers)
    float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ssc = bsc, ssh = bsh;
    if (XFA) {
        const float4 g = *reinterpret_cast<const float4*>(a.xfb.scale + cb0 + cq * 4);
        bsc = make_float4(g.x * a.xfb.mult, g.y * a.xfb.mult, g.z * a.xfb.mult, g.w * a.xfb.mult);
        bsh = *reinterpret_cast<const float4*>(a.xfb.shift + cb0 + cq * 4);
    }
    if (XFS) {
        const float4 g = *reinterpret_cast<const float4*>(a.xfs.scale + cs0 + csq * 4);
        ssc = make_float4(g.x * a.xfs.mult, g.y * a.xfs.mult, g.z * a.xfs.mult, g.w * a.xfs.mult);
        ssh = *reinterpret_cast<const float4*>(a.xfs.shift + cs0 + csq * 4);
    }
    const float balpha = a.xfb.alpha, salpha = a.xfs.alpha;
    // Pattern-word form: element = (dxhat * w_f[ch]) * (bit ch ? scale : scale * alpha).  (Round 6 also tried the four selects of a channel quad as ONE
    // 16-byte LDS table row indexed by the quad's nibble of the word -- 133 fewer VALU instructions per tile and wave, same bits -- and measured it
    // SLOWER: dec3.wgrad 0.86 -> 0.89-0.91 of the round-5 kernel's time, with or without bank padding: profiles/r06_b_fb_nibble_table_ab.md.)
    float fb_wf[4] = {0.f, 0.f, 0.f, 0.f}, fb_s1[4] = {0.f, 0.f, 0.f, 0.f}, fb_s0[4] = {0.f, 0.f, 0.f, 0.f};
    if (FBB) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            fb_wf[e] = a.xfb.fb_wf[cb0 + cq * 4 + e];
            fb_s1[e] = a.xfb.scale[cb0 + cq * 4 + e] * a.xfb.mult;
            fb_s0[e] = fb_s1[e] * a.xfb.alpha;
        }
    }
    const unsigned fb_shift = (unsigned)(cb0 + cq * 4);

    constexpr int MAXT = 7;
    // fragment addressing (ushort indices): group g = lane >> 4 -> channel half g & 1, position half lh = g >> 1; lane i of the group -> row i >> 2
    // of the 4-position block, columns 4 (i & 3)
    const int li = lane & 15, chalf = (lane >> 4) & 1;
    const int a_base = ((2 * lh) * IW + 2 * (li >> 2)) * LDH + 16 * chalf + 4 * (li & 3);      // + (4 js IW + ky IW + kx + 8 t) LDH
    const int b_base = (8 * lh + (li >> 2)) * LDQ + csb * 32 + 16 * chalf + 4 * (li & 3);        // + (16 js + 4 t) LDQ

    v16f acc[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    auto tr8 = [&](const unsigned short* plane, const int addr, const int step) __attribute__((always_inline)) {      // 8 consecutive k of this lane's channel
        const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(const_cast<unsigned short*>(plane) + addr));
        const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(const_cast<unsigned short*>(plane) + addr + step));
        const uint2 lo2 = __builtin_bit_cast(uint2, r0), hi2 = __builtin_bit_cast(uint2, r1);
        return make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
    };
    struct Pl { uint4 p[NPL]; };
    auto trp = [&](const unsigned short* planes, const int plane_stride, const int addr, const int step) __attribute__((always_inline)) {
        Pl r;
#pragma unroll
        for (int p = 0; p < NPL; ++p) r.p[p] = tr8(planes + p * plane_stride, addr, step);
        return r;
    };
    auto mma3 = [&](v16f& c, const Pl& a_, const Pl& b_) __attribute__((always_inline)) {
        c = mfma_bf16(a_.p[0], b_.p[0], c);
        c = mfma_bf16(a_.p[0], b_.p[1], c);
        c = mfma_bf16(a_.p[1], b_.p[0], c);
        if constexpr (NPL == 3) {      // bf16x6: + a0 b2, a2 b0, a1 b1
            c = mfma_bf16(a_.p[0], b_.p[2], c);
            c = mfma_bf16(a_.p[2], b_.p[0], c);
            c = mfma_bf16(a_.p[1], b_.p[1], c);
        }
    };

    // ---- tile loop, software-pipelined: the global loads of tile t + 1 are in flight while tile t is contracted ----
    uint4 v[FBB ? 1 : PER];
    unsigned vb[FBB ? PER : 1];
    float vg[FBB ? PER : 1];
    uint4 sv[SPER];
    // coordinates of the NEXT tile to be issued (wave-uniform, advanced incrementally)
    int nx_tx0 = (t_begin % tilesx) * TW, nx_ty0 = ((t_begin / tilesx) % tilesy) * TH, nx_n = t_begin / (tilesx * tilesy);
    unsigned bad = 0;            // bit u: element u of the tile in flight is outside the image
    const size_t big_img = (size_t)d.HB * d.WB * (FBB ? 1 : d.CB);          // elements of one sample of the big operand (FBB: of the per-pixel words)
    const size_t origin_back = (size_t)(d.WB + 1) * (FBB ? 1 : d.CB);      // the halo's origin is one row and one column before the tile's first pixel
    auto issue = [&]() __attribute__((always_inline)) {
        const int tx0 = nx_tx0, ty0 = nx_ty0, n = nx_n;
        bad = ((ty0 == 0) ? clsT : 0u) | ((ty0 + TH == d.HS) ? clsB : 0u) | ((tx0 == 0) ? clsL : 0u) | ((tx0 + TW == d.WS) ? clsR : 0u);
        const unsigned tile_b = (unsigned)((2 * ty0) * d.WB + 2 * tx0) * pixbytes;
        if (FBB) {
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<unsigned*>(a.xfb.fb_bits) + (size_t)n * big_img - origin_back), 0, 0x80000000u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<float*>(a.xfb.fb_dxhat) + (size_t)n * big_img - origin_back), 0, 0x80000000u, 0x00020000);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const unsigned off = (bad >> u) & 1u ? 0x80000000u : voff[u];
                vb[FBB ? u : 0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, (int)off, (int)tile_b, 0);
                vg[FBB ? u : 0] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rg, (int)off, (int)tile_b, 0));
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<float*>(a.big) + (size_t)n * big_img - origin_back), 0, 0x80000000u, 0x00020000);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const unsigned off = (bad >> u) & 1u ? 0x80000000u : voff[u];
                v[FBB ? 0 : u] = buf_load16(rs, off, tile_b);
            }
        }
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(const_cast<float*>(a.small_) + ((size_t)(n * d.HS + ty0) * d.WS + tx0) * d.CS), 0, 0x80000000u, 0x00020000);
#pragma unroll
        for (int u = 0; u < SPER; ++u) sv[u] = buf_load16(rq, svoff[u], 0);
        nx_tx0 += TW;
        if (nx_tx0 == d.WS) { nx_tx0 = 0; nx_ty0 += TH; if (nx_ty0 == d.HS) { nx_ty0 = 0; ++nx_n; } }
    };
    auto as_f4 = [](uint4 q) { return make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)); };
    auto lds_store8 = [&](unsigned byte_addr, uint2 q) __attribute__((always_inline)) { *reinterpret_cast<uint2*>(dsm + byte_addr) = q; };
    auto commit = [&](const unsigned bufb) __attribute__((always_inline)) {      // bufb: byte offset of the buffer written
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            float4 tv;
            if (FBB) {
#pragma clang fp contract(off)      // (the products are ROUNDED before the hi | lo split subtracts from them, as in round 5 where a select stood between the two)
                const unsigned b = vb[FBB ? u : 0] >> fb_shift;
                const float gq = vg[FBB ? u : 0];      // 0 outside the image (descriptor range check): the products below are zeros
                tv.x = gq * fb_wf[0] * ((b & 1u) ? fb_s1[0] : fb_s0[0]);
                tv.y = gq * fb_wf[1] * ((b & 2u) ? fb_s1[1] : fb_s0[1]);
                tv.z = gq * fb_wf[2] * ((b & 4u) ? fb_s1[2] : fb_s0[2]);
                tv.w = gq * fb_wf[3] * ((b & 8u) ? fb_s1[3] : fb_s0[3]);
            } else {
                tv = as_f4(v[FBB ? 0 : u]);
                if (XFA) tv = keep4(!((bad >> u) & 1u), xform4(tv, bsc, bsh, balpha));      // padding is zero AFTER the activation
            }
            uint2 pl[NPL];
            split_planes<NPL>(tv, pl);
            const unsigned o = (u + 1 < PER) ? lds_b + (unsigned)(u * DP * LDH) * 2u : lds_last;
#pragma unroll
            for (int p = 0; p < NPL; ++p) lds_store8(bufb + o + (unsigned)(p * BIGP) * 2u, pl[p]);
        }
#pragma unroll
        for (int u = 0; u < SPER; ++u) {
            float4 tv = as_f4(sv[u]);
            if (XFS) tv = xform4(tv, ssc, ssh, salpha);
            uint2 pl[NPL];
            split_planes<NPL>(tv, pl);
            const unsigned o = (unsigned)(NPL * BIGP) * 2u + slds_b + (unsigned)(u * (NT / CSQ) * LDQ) * 2u;
#pragma unroll
            for (int p = 0; p < NPL; ++p) lds_store8(bufb + o + (unsigned)(p * SMALLP) * 2u, pl[p]);
        }
    };
    if (t_begin < t_end) {
        issue();
        if (DB) { commit(0u); if (t_begin + 1 < t_end) issue(); __syncthreads(); }
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = DB ? ((t - t_begin) & 1) * BUFU : 0;      // ushort offset of the buffer read
        if (!DB) {
            __syncthreads();                   // the previous tile's fragments are consumed
            commit(0u);
            if (t + 1 < t_end) issue();        // in flight across the barrier and the MFMA loop below
            __syncthreads();
        }
        const unsigned short* bP = bPl + cur;
        const unsigned short* sP = sPlT + cur;
#pragma unroll
        for (int js = 0; js < TH * TW / 16; ++js) {
            // positions 16 js .. 16 js + 15 = tile rows 2 js (k half 0) and 2 js + 1 (k half 1)
            const int bo = b_base + 16 * js * LDQ;
            const Pl bq_ = trp(sP, SMALLP, bo, 4 * LDQ);
            const int ao = a_base + (4 * js * IW + wave * IW) * LDH;          // this wave's kernel row
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) mma3(acc[kx], trp(bP, BIGP, ao + kx * LDH, 8 * LDH), bq_);
            const int a4 = a_base + (4 * js * IW + 4 * IW) * LDH;             // kernel row 4
            mma3(acc[5], trp(bP, BIGP, a4 + wave * LDH, 8 * LDH), bq_);        // tap (4, wave)
            if (DB && js == 1) {
                // tile t + 1 (in registers since tile t - 1's MFMAs) -> the other buffer; then the loads of tile t + 2
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < t_end) { commit((unsigned)(BUFU - cur) * 2u); if (t + 2 < t_end) issue(); }      // (unconditional, clamped forms measured the same)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            // this wave's quarter of tap (4, 4) = position block js = wave, addressed by a run-time offset BEHIND the straight-line loop: with the
            // `if (js == wave)` inside it (round 4) every js step was its own basic block and the fragment reads of a block could not be hoisted
            // above the previous block's MFMAs.  acc[6] receives the same three products in the same order.
            const int bo = b_base + 16 * wave * LDQ;
            const int a4 = a_base + (4 * wave * IW + 4 * IW) * LDH;
            mma3(acc[6], trp(bP, BIGP, a4 + 4 * LDH, 8 * LDH), trp(sP, SMALLP, bo, 4 * LDQ));
        }
        if (DB) __syncthreads();          // tile t + 1 is complete in LDS, and every wave is done with tile t's buffer
    }

    // Accumulator block -> slab.  One 64-bit address per lane; everything else is a wave-uniform 32-bit offset (the plain form
    // `out[((size_t)tap * CB + cb) * CS + col]` cost ~400 quarter-rate 64-bit multiply-adds per wave: 8 of this kernel's ~47 us).
    const int CSi = d.CS;
    float* ob = a.partial + (size_t)blockIdx.z * a.Mtot * CSi + (size_t)(cb0 + 4 * lh) * CSi + cs0 + csb * 32 + l31;
    const int tapstride = d.CB * CSi;
    if (a.abl & 32) {       // stress test (tests/test_gpu_knobs.py): every workgroup idles ~100 us before it writes its slab, so that the filter gradient
                            // OUTLASTS the any-order data gradient launched behind it -- results stay correct, only the timing changes
        for (int i = 0; i < 32; ++i) __builtin_amdgcn_s_sleep(127);
    }
#pragma unroll
    for (int j = 0; j < MAXT - 1; ++j) {
        const int tap = j < 5 ? 5 * wave + j : 20 + wave;      // (row, 0..4), (4, row)
        float* ot = ob + tap * tapstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2)) * CSi] = acc[j][r];
    }
    // tap (4, 4): the four rows' shares are folded through LDS in a fixed order
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[((csb * 4 + wave) * 16 + r) * 64 + lane] = acc[MAXT - 1][r];
    __syncthreads();
    if (wave == 0) {
        float* ot = ob + 24 * tapstride;
        const float* q = sRed + (size_t)csb * 64 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ot[((r & 3) + 8 * (r >> 2)) * CSi] = (q[r * 64 + lane] + q[(16 + r) * 64 + lane]) + (q[(32 + r) * 64 + lane] + q[(48 + r) * 64 + lane]);
    }
    if (a.dbgbuf && tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.dbgbuf[2 * b] = dbg_t0;
        a.dbgbuf[2 * b + 1] = wall_clock64();
    }
}

constexpr size_t conv5_w_bf16_tr_lds_bytes(int ncsb, int npl = 2) { return (size_t)npl * 19 * 19 * 40 * 2 + (size_t)npl * 64 * (32 * ncsb + 8) * 2 + 192 * 4; }

#include "uad_convk16.inc"

struct W5Choice { bool ok; int splits, tiles_per_split, total_tiles; };
inline W5Choice choose_w5(const UadConvDesc& d, bool bf16 = true) {
    W5Choice c{false, 0, 0, 0};
    if (!(d.KS == 5 && d.S == 2 && d.P == 1 && d.HB == 2 * d.HS && d.WB == 2 * d.WS)) return c;
    if (d.CB % 32 || d.CS % 32 || d.HS % 8 || d.WS % 8) return c;
    c.total_tiles = d.N * (d.HS / 8) * (d.WS / 8);
    const int blocks = (d.CB / 32) * (d.CS / 32);
    // UAD_W5_TARGET (tuning knob): workgroup-slab target of a launch.  Every split costs one [25][CB][CS] slab written and re-read, every tile
    // ~2.6 us of a workgroup's life.  Default 448 (round 4, profiles/r04_i_w5_target_sweep.log): the kernel itself is fastest at 512 (two
    // workgroups on every CU: dec3.wgrad 61 us, 68 at 448), but the layer's data gradient is launched right behind it without the barrier bit and
    // fills the slots a 410-workgroup filter gradient leaves free -- the STEP is 2.5 % faster (0.916 -> 0.894 ms), and the slabs are a fifth smaller.
    // The exact-fp32 kernel (bf16 = false) is bound by the matrix pipe, not by latency: it keeps 512 (448 cost that mode 4 %: 1.704 -> 1.772 ms).
    static const int target_env = getenv("UAD_W5_TARGET") ? atoi(getenv("UAD_W5_TARGET")) : 0;
    // Round 6, double-buffered kernel: re-swept on two boxes (profiles/r06_i_w5_target_sweep_db.log) -- 384 is 1 % faster over the step than 448 in all four pairs
    // (0.822 / 0.830 vs 0.833 / 0.836 ms; 0.848 / 0.841 vs 0.856 / 0.852), although the kernels' own tags sum to 260 us instead of 250.
    const int target = target_env > 0 ? target_env : (bf16 ? 384 : 512);
    int splits = (target + blocks - 1) / blocks;
    if (splits > c.total_tiles) splits = c.total_tiles;
    if (splits < 1) splits = 1;
    c.tiles_per_split = (c.total_tiles + splits - 1) / splits;
    c.splits = (c.total_tiles + c.tiles_per_split - 1) / c.tiles_per_split;
    c.ok = true;
    return c;
}

inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct TileChoice { int BM, BN, BK; };

inline TileChoice choose_tile(long M, int Nn, int CA, int classes) {
    TileChoice t;
    t.BK = (CA % 32 == 0) ? 32 : (CA % 16 == 0) ? 16 : 8;
    t.BN = (Nn > 32) ? 64 : 32;
    auto nwg = [&](int bm) { return ((M + bm - 1) / bm) * ((Nn + t.BN - 1) / t.BN) * classes; };
    t.BM = (t.BK == 32 && nwg(128) >= 512) ? 128 : 64;
    return t;
}

// split-K factor of the generic kernel: only when the plain grid cannot fill the chip and K is long enough
inline int choose_nsplit(long M, int Nn, int CA, int classes, int min_taps) {
    const TileChoice t = choose_tile(M, Nn, CA, classes);
    const long wgs = ((M + t.BM - 1) / t.BM) * ((Nn + t.BN - 1) / t.BN) * classes;
    const int nk_min = min_taps * (CA / t.BK);
    if (wgs >= 256 || nk_min < 8 || (Nn % 4)) return 1;
    long s = (512 + wgs - 1) / wgs;
    if (s > nk_min / 4) s = nk_min / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

template <int KIND>
void launch_gemm(const ConvGemmArgs& a, int classes, hipStream_t st) {
    const TileChoice t = choose_tile(a.M, a.Nn, a.CA, classes);
    dim3 grid((a.M + t.BM - 1) / t.BM, (a.Nn + t.BN - 1) / t.BN, classes * a.nsplit);
    if (a.math16 && !a.xf.scale && t.BK == 32 && t.BN == 64 && !getenv("UAD_NO_G16")) {
        if (t.BM == 128) { hipLaunchKernelGGL((conv_gemm16_kernel<128, 64, 32, 2, 2, KIND>), grid, dim3(256), 0, st, a); return; }
        if (t.BM == 64) { hipLaunchKernelGGL((conv_gemm16_kernel<64, 64, 32, 2, 2, KIND>), grid, dim3(256), 0, st, a); return; }
    }
#define UAD_GEMM_CASE(bm, bn, bk, wgm, wgn)                                                              \
    if (t.BM == bm && t.BN == bn && t.BK == bk) {                                                        \
        hipLaunchKernelGGL((conv_gemm_kernel<bm, bn, bk, wgm, wgn, KIND>), grid, dim3(64 * wgm * wgn), 0, st, a); \
        return;                                                                                          \
    }
    UAD_GEMM_CASE(128, 64, 32, 2, 2)
    UAD_GEMM_CASE(128, 32, 32, 4, 1)
    UAD_GEMM_CASE(64, 64, 32, 2, 2)
    UAD_GEMM_CASE(64, 32, 32, 2, 1)
    UAD_GEMM_CASE(64, 64, 16, 2, 2)
    UAD_GEMM_CASE(64, 32, 16, 2, 1)
    UAD_GEMM_CASE(64, 64, 8, 2, 2)
    UAD_GEMM_CASE(64, 32, 8, 2, 1)
#undef UAD_GEMM_CASE
}

}  // namespace

namespace {
enum { PATH_GENERIC = 0, PATH_SPATIAL = 1, PATH_SPLITK = 2 };
struct GemmPlan { int path; SpatialChoice sc; int nsplit; int tiles; size_t ws_floats; long long out_elems; int out_rows; bool inkernel; };

// One decision procedure for launchers and for the colpart-tile / workspace queries.
// ncounters > 0: the caller runs the split-bf16 spatial kernels and owns that many arrival counters -> a split launch reduces its slabs in the
// kernel (last workgroup per tile) when the instance that will run supports it
inline GemmPlan plan_gemm(const UadConvDesc& d, bool f_type, bool have_pack, size_t ws_cap, int ncounters = 0) {
    GemmPlan p;
    p.inkernel = false;
    const int CA = f_type ? d.CB : d.CS, Nn = f_type ? d.CS : d.CB;
    const long M = (long)d.N * d.HS * d.WS;
    const int classes = f_type ? 1 : d.S * d.S;
    int min_taps = d.KS * d.KS;
    if (!f_type && d.S > 1) { const int t = d.KS / d.S; min_taps = t * t; }
    p.out_rows = f_type ? (int)M : d.N * d.HB * d.WB;
    p.out_elems = (long long)p.out_rows * Nn;
    p.sc = have_pack ? choose_spatial(d, CA, Nn, f_type) : SpatialChoice{false, 0, 0, 0, 0};
    p.nsplit = 1; p.ws_floats = 0;
    const TileChoice t = choose_tile(M, Nn, CA, classes);
    int ns = choose_nsplit(M, Nn, CA, classes, min_taps);
    if (ns > 1 && (size_t)ns * p.out_elems > ws_cap) ns = 1;
    if (p.sc.ok) {
        // the spatial kernels want >= 2 workgroups per CU (staging / epilogue of one overlaps the MFMA work of the other);
        // if the plain grid is smaller, split the channel chunks over workgroups (slabs + splitk_epilogue_kernel)
        const long wgs = (long)d.N * (d.HS / p.sc.TH) * (d.WS / p.sc.TW) * (Nn / p.sc.BN);
        const int chunks = CA / p.sc.CK;
        int sp = 1;
        // Workgroups a spatial launch is split up to.  Round 4, same-box sweep (profiles/r04_d_split_target_sweep.log): the gather (F) kernels
        // want two workgroups per CU (enc3.fwd 22 us at 512 / 23 at 256 / 30 unsplit); the class-sequential scatter (D) kernels stage their whole
        // channel range once and pay the slab exchange four times (once per parity class): they are faster with HALF the splits (dec0.fwd 30 -> 25,
        // dec1.fwd 37 -> 31, enc3.dgrad 35 -> 31, enc2.dgrad 38 -> 32 us).  UAD_SPLIT_TARGET=n overrides both (tuning knob).
        static const int split_env = getenv("UAD_SPLIT_TARGET") ? atoi(getenv("UAD_SPLIT_TARGET")) : 0;
        const int split_target = split_env > 0 ? split_env : (f_type ? 512 : 256);
        while (wgs * sp < split_target && sp * 2 <= chunks && (size_t)(sp * 2) * p.out_elems <= ws_cap) sp *= 2;
        // Fewest workgroups a spatial launch may have; below it the launch falls back to the generic implicit-GEMM kernel (+ a separate split-K epilogue
        // launch).  Round 5: 128 (was 256): at the 16-slice workloads of BASELINE configs[2] / [4] the 8 x 8 layers come to 128 workgroups after
        // splitting, and the spatial kernel with its in-kernel slab reduction beats generic + epilogue by 2x there (spatial-GMVAE restoration:
        // enc4.fwd 40 -> 16 us, dec0.dgrad 41 -> 17, the iteration 0.699 -> 0.632 ms; profiles/r05_e_planner_min_wgs.log).  UAD_SPATIAL_MIN_WGS=n overrides.
        static const int min_env = getenv("UAD_SPATIAL_MIN_WGS") ? atoi(getenv("UAD_SPATIAL_MIN_WGS")) : 0;
        // (only for SPLIT launches whose slabs are reduced inside the kernel -- bf16x3 handles with ticket counters: that is the case measured; unsplit
        // launches and the exact-fp32 kernels, which would still need the separate epilogue launch, keep 256)
        const int min_wgs = min_env > 0 ? min_env : ((ncounters > 0 && sp > 1) ? 128 : 256);
        if (wgs * sp >= (split_target < min_wgs ? split_target : min_wgs)) {
            p.path = PATH_SPATIAL;
            p.nsplit = sp;
            p.ws_floats = sp > 1 ? (size_t)sp * p.out_elems : 0;
            if (sp > 1 && ncounters > 0 && wgs <= ncounters && (size_t)sp * p.out_elems * 4 < ((size_t)1 << 32)) {
                static const bool off = getenv("UAD_NO_INKERNEL_SPLITK") != nullptr;
                const int cst = CA / sp;
                const bool inst = f_type || (CA % sp == 0 && (cst == 32 || cst == 64 || (cst == 128 && p.sc.BN == 64)));
                p.inkernel = !off && inst;
            }
            p.tiles = (sp > 1 && !p.inkernel) ? (p.out_rows + 63) / 64 : d.N * (d.HS / p.sc.TH) * (d.WS / p.sc.TW);
            return p;
        }
    }
    if (ns > 1) {
        p.path = PATH_SPLITK; p.nsplit = ns; p.ws_floats = (size_t)ns * p.out_elems;
        p.tiles = (p.out_rows + 63) / 64;
        return p;
    }
    p.path = PATH_GENERIC;
    p.tiles = (int)((M + t.BM - 1) / t.BM) * classes;
    return p;
}
}  // namespace

bool uad_conv_k3_takes(const UadConvDesc& d, bool f_type) { return convk16_shape_ok(d, f_type); }
void uad_k3_prof_enable(bool on) {
    if (on && !g_k3prof) { for (auto& r : g_k3recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } g_k3recs.clear(); }
    g_k3prof = on;
}
// one text line per launch shape: "kind p1 p2 ntaps npl N MH MW CA Nn calls total_ms form"; returns the number of bytes the full table needs
int uad_k3_prof_read(char* buf, int cap) {
    (void)hipDeviceSynchronize();
    std::map<std::array<int, 11>, std::pair<int, double>> agg;
    for (auto& r : g_k3recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        std::array<int, 11> k;
        for (int i = 0; i < 11; ++i) k[i] = r.key[i];
        auto& e = agg[k];
        e.first += 1; e.second += ms;
    }
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        int n = 0;
        for (int i = 0; i < 10; ++i) n += snprintf(line + n, sizeof line - n, "%d ", kv.first[i]);
        snprintf(line + n, sizeof line - n, "%d %.6f %d\n", kv.second.first, kv.second.second, kv.first[10]);
        out += line;
    }
    if (buf && cap > 0) { const size_t c = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1; memcpy(buf, out.data(), c); buf[c] = 0; }
    return (int)out.size() + 1;
}
bool uad_conv_spatial_ok(const UadConvDesc& d, bool f_type) {
    if (convk16_shape_ok(d, f_type)) return true;      // k3 tap-list kernel (bf16x3 planes only: the fp32 pack of such a tensor is simply not used)
    return f_type ? choose_spatial(d, d.CB, d.CS, true).ok : choose_spatial(d, d.CS, d.CB, false).ok;
}
namespace {
// bf16x6 (three-plane) launches run on the F-kind and the lane = pixel spatial kernels; every other launch of a handle in that mode takes the exact-fp32
// route (its fp32 pack is kept beside the planes).  One decision for the launchers and the tile queries.
bool x6_route(const UadConvDesc& d, bool f_type, const GemmPlan& p) {
    if (p.path != PATH_SPATIAL) return false;
    if (f_type) return true;       // (final-backward-on-load forms: the 32-column instance only -- they have CA <= 32, which the planner gives that instance)
    if (d.CB % 32) return false;
    if (p.nsplit > 1 && !p.inkernel) return false;
    const int cst = (d.CS % p.nsplit == 0) ? d.CS / p.nsplit : 0;
    return p.sc.BN == 64 ? (cst == 32 || cst == 64 || cst == 128) : (cst == 32 || cst == 64);
}
GemmPlan plan_for(const UadConvDesc& d, bool f_type, bool have_pack, size_t ws_floats, int ncounters, int planes16) {
    GemmPlan p = plan_gemm(d, f_type, have_pack, ws_floats, ncounters);
    if (planes16 == 3 && !x6_route(d, f_type, p)) p = plan_gemm(d, f_type, have_pack, ws_floats, 0);      // the exact-fp32 kernels: no in-kernel slab reduction
    return p;
}
}  // namespace
bool uad_conv_x6_takes(const UadConvDesc& d, bool f_type, size_t ws_floats, int ncounters) { return x6_route(d, f_type, plan_gemm(d, f_type, true, ws_floats, ncounters)); }
int uad_conv_f_tiles(const UadConvDesc& d, bool have_pack, size_t ws_floats, int ncounters, int planes16) { return plan_for(d, true, have_pack, ws_floats, ncounters, planes16).tiles; }
int uad_conv_d_tiles(const UadConvDesc& d, bool have_pack, size_t ws_floats, int ncounters, int planes16) { return plan_for(d, false, have_pack, ws_floats, ncounters, planes16).tiles; }
bool uad_conv_f_supports_final_bwd(const UadConvDesc& d, bool have_pack16, size_t ws_floats) {
    if (!have_pack16) return false;
    const GemmPlan p = plan_gemm(d, true, true, ws_floats);
    return p.path == PATH_SPATIAL;
}
bool uad_conv_d_can_fuse_final(const UadConvDesc& d, bool have_pack16, size_t ws_floats) {
    if (!have_pack16 || getenv("UAD_NO_FUSED_FINAL")) return false;
    const GemmPlan p = plan_gemm(d, false, true, ws_floats);
    // conv5_d16_kernel<8,16,CST,4,1> with the whole channel range in one workgroup column block
    return p.path == PATH_SPATIAL && p.nsplit == 1 && p.sc.BN == 32 && d.CB == 32 && (d.CS == 32 || d.CS == 64);
}
// exact-fp32 mode: conv5_d_kernel<8,16,32,4,1,FIN> (UAD_NO_FUSED_FINAL_F32=1: the separate final_kernel pass)
bool uad_conv_d_can_fuse_final_f32(const UadConvDesc& d, bool have_pack, size_t ws_floats) {
    if (!have_pack || getenv("UAD_NO_FUSED_FINAL") || getenv("UAD_NO_FUSED_FINAL_F32")) return false;
    const GemmPlan p = plan_gemm(d, false, true, ws_floats);
    return p.path == PATH_SPATIAL && p.nsplit == 1 && p.sc.TH == 8 && p.sc.TW == 16 && p.sc.BN == 32 && p.sc.CK == 32 && d.CB == 32;
}
size_t uad_conv_ws_floats(const UadConvDesc& d, bool f_type, bool have_pack) { return plan_gemm(d, f_type, have_pack, (size_t)1 << 40).ws_floats; }

void uad_launch_pack_weights(const float* params, float* wpack_f, float* wpack_d, const long long* offs, const int* cbs,
                             const int* css, const int* taps, int n, hipStream_t st) {
    PackDesc pd;
    long long total = 0;
    pd.n = n;
    for (int i = 0; i < n; ++i) {
        pd.off[i] = offs[i]; pd.cb[i] = cbs[i]; pd.cs[i] = css[i]; pd.count[i] = taps[i] * cbs[i] * css[i];
        total += pd.count[i];
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, params, wpack_f, wpack_d, pd);
}

void uad_launch_pack_weights_bf16_3p(const float* params, unsigned short* w3_f, unsigned short* w3_d, const long long* offs,
                                     const int* cbs, const int* css, const int* taps, int n, hipStream_t st) {
    PackDesc pd;
    pd.n = n;
    long long total = 0;
    for (int i = 0; i < n; ++i) { pd.off[i] = offs[i]; pd.cb[i] = cbs[i]; pd.cs[i] = css[i]; pd.count[i] = taps[i] * cbs[i] * css[i]; total += pd.count[i]; }
    bool v8 = ((((uintptr_t)params | (uintptr_t)w3_f | (uintptr_t)w3_d) & 15) == 0);
    for (int i = 0; i < n; ++i) v8 = v8 && pd.count[i] % 8 == 0 && pd.cb[i] % 8 == 0 && pd.cs[i] % 8 == 0;      // whole 8-element groups (any dword-aligned offset: f4u / u4u)
    if (v8) hipLaunchKernelGGL(pack_weights_bf16_v8_kernel<3>, dim3((unsigned)((total / 8 + 255) / 256), 2), dim3(256), 0, st, params, w3_f, w3_d, pd);
    else hipLaunchKernelGGL(pack_weights_bf16_3p_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, params, w3_f, w3_d, pd);
}

namespace {
void run_plan(const GemmPlan& p, ConvGemmArgs& a, bool f_type, float* ws, hipStream_t st) {
    const UadConvDesc& d = a.d;
    a.nsplit = 1; a.out_elems = p.out_elems;
    a.dbg = 0; a.dbgbuf = nullptr;
    if (p.path == PATH_SPATIAL) {
        float* out = a.Out;
        if (p.nsplit > 1) { a.Out = ws; a.nsplit = p.nsplit; a.out_final = out; }
        if (!(p.nsplit > 1 && p.inkernel && a.Wp16)) a.sk_counter = nullptr;
        dim3 grid((d.HS / p.sc.TH) * (d.WS / p.sc.TW), d.N, (a.Nn / p.sc.BN) * p.nsplit);
        if (a.Wp16) {   // bf16x3 math mode
            if (f_type) {
                if (a.npl == 3) {      // bf16x6: 16-channel chunks in the 64-column instance too (three planes of a 32-channel 19 x 19 halo = 87 KB: one workgroup per CU)
                    if (p.sc.BN == 64) {
                        if (a.xf.fb_bits || a.xf.fb_dxhat) { fprintf(stderr, "uad: final-backward on load runs on the 32-column instance (CA <= 32)\n"); abort(); }
                        if (a.xf.scale) launch_conv5_f16_v<8, 8, 16, 2, 2, 0, true, 3>(a, grid, st); else launch_conv5_f16_v<8, 8, 16, 2, 2, 0, false, 3>(a, grid, st);
                    } else launch_conv5_f16<8, 16, 16, 4, 1, 3>(a, grid, st);
                } else
                if (p.sc.BN == 64) launch_conv5_f16<8, 8, 32, 2, 2>(a, grid, st);
                else launch_conv5_f16<8, 16, 16, 4, 1>(a, grid, st);
            } else {
                // class-sequential kernel whenever the workgroup's whole channel range fits in LDS (CA == cst * nsplit)
                const int cst = (a.CA % p.nsplit == 0) ? a.CA / p.nsplit : 0;
                constexpr bool seq = true;      // class-sequential generation (conv5_d16_kernel): the fallback of the lane = pixel one
                const bool lp = seq && conv5_d16s_takes(a);      // lane = pixel generation (uad_conv16s.inc)
                if (lp && ((p.sc.BN == 64 && (cst == 128 || cst == 64 || cst == 32)) || (p.sc.BN == 32 && (cst == 64 || cst == 32)))) launch_conv5_d16s_any(a, grid, p.sc.BN, cst, st);
                else if (a.npl == 3) { fprintf(stderr, "uad: a bf16x6 launch reached a kernel without the three-plane form (x6_route decides before run_plan)\n"); abort(); }
                else
                if (p.sc.BN == 64) {
                    if (seq && cst == 128) launch_conv5_d16<8, 8, 128, 2, 2>(a, grid, st);
                    else if (seq && cst == 64) launch_conv5_d16<8, 8, 64, 2, 2>(a, grid, st);
                    else if (seq && cst == 32) launch_conv5_d16<8, 8, 32, 2, 2>(a, grid, st);
                    else launch_conv5_bf16<8, 8, 32, 2, 2, KIND_D>(a, grid, st);
                } else if (seq && cst == 64) launch_conv5_d16<8, 16, 64, 4, 1>(a, grid, st);
                else if (seq && cst == 32) launch_conv5_d16<8, 16, 32, 4, 1>(a, grid, st);
                else if (p.sc.CK == 32) launch_conv5_bf16<8, 16, 32, 4, 1, KIND_D>(a, grid, st);
                else launch_conv5_bf16<8, 16, 16, 4, 1, KIND_D>(a, grid, st);
            }
        } else if (f_type) {
            if (p.sc.BN == 64) UAD_SPATIAL_LAUNCH((conv5_f_kernel<8, 8, 32, 2, 2>), grid, dim3(256), 0, st, a);
            else UAD_SPATIAL_LAUNCH((conv5_f_kernel<8, 16, 16, 4, 1>), grid, dim3(256), 0, st, a);
        } else {
            if (p.sc.BN == 64 && a.ep.kind != UAD_EPI_FINAL) UAD_SPATIAL_LAUNCH((conv5_d_kernel<8, 8, 32, 2, 2>), grid, dim3(256), 0, st, a);
            else if (p.sc.CK == 32 && a.ep.kind == UAD_EPI_FINAL && p.nsplit == 1 && a.Nn == 32) UAD_SPATIAL_LAUNCH((conv5_d_kernel<8, 16, 32, 4, 1, true>), grid, dim3(256), 0, st, a);
            else if (a.ep.kind == UAD_EPI_FINAL) { fprintf(stderr, "uad: fused final epilogue asked of an exact-fp32 launch that cannot run it (check uad_conv_d_can_fuse_final_f32 first)\n"); abort(); }
            else if (p.sc.CK == 32) UAD_SPATIAL_LAUNCH((conv5_d_kernel<8, 16, 32, 4, 1>), grid, dim3(256), 0, st, a);
            else UAD_SPATIAL_LAUNCH((conv5_d_kernel<8, 16, 16, 4, 1>), grid, dim3(256), 0, st, a);
        }
        if (p.nsplit > 1 && !a.sk_counter) {
            dim3 g2((p.out_rows + 63) / 64, (a.Nn + 63) / 64);
            hipLaunchKernelGGL(splitk_epilogue_kernel, g2, dim3(256), 0, st, ws, p.nsplit, p.out_elems, p.out_rows, a.Nn, a.ep, out);
        }
        return;
    }
    const int classes = f_type ? 1 : d.S * d.S;
    if (p.path == PATH_SPLITK) {
        float* out = a.Out;
        a.Out = ws; a.nsplit = p.nsplit;
        if (f_type) launch_gemm<KIND_F>(a, classes, st); else launch_gemm<KIND_D>(a, classes, st);
        dim3 grid((p.out_rows + 63) / 64, (a.Nn + 63) / 64);
        hipLaunchKernelGGL(splitk_epilogue_kernel, grid, dim3(256), 0, st, ws, p.nsplit, p.out_elems, p.out_rows, a.Nn, a.ep, out);
        return;
    }
    if (f_type) launch_gemm<KIND_F>(a, classes, st); else launch_gemm<KIND_D>(a, classes, st);
}
}  // namespace

void uad_launch_pack_weights_bf16(const float* params, unsigned short* w16_f, unsigned short* w16_d, const long long* offs,
                                  const int* cbs, const int* css, const int* taps, int n, hipStream_t st) {
    PackDesc pd;
    long long total = 0;
    pd.n = n;
    for (int i = 0; i < n; ++i) {
        pd.off[i] = offs[i]; pd.cb[i] = cbs[i]; pd.cs[i] = css[i]; pd.count[i] = taps[i] * cbs[i] * css[i];
        total += pd.count[i];
    }
    bool v8 = ((((uintptr_t)params | (uintptr_t)w16_f | (uintptr_t)w16_d) & 15) == 0);
    for (int i = 0; i < n; ++i) v8 = v8 && pd.cb[i] % 8 == 0 && pd.cs[i] % 8 == 0;      // (any dword-aligned offset: f4u / u4u)
    if (v8) hipLaunchKernelGGL(pack_weights_bf16_v8_kernel<2>, dim3((unsigned)((total / 8 + 255) / 256), 2), dim3(256), 0, st, params, w16_f, w16_d, pd);
    else hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, params, w16_f, w16_d, pd);
}

void uad_launch_conv_f(const UadConvDesc& d, const float* big_in, UadXform xf, const float* W, float* small_out,
                       UadEpilogue ep, hipStream_t st, const float* Wpacked, UadGemmWs ws, const unsigned short* Wp16,
                       long long w16_plane, bool generic_bf16x3, int planes16) {
    if (convk16_takes(d, true, xf, ep, Wp16)) { run_convk16(d, true, big_in, small_out, ep, Wp16, w16_plane, planes16, st); return; }      // k3 s1 / s2, bf16x3 / bf16x6
    ConvGemmArgs a;
    a.Wp = Wpacked; a.Wp16 = Wp16; a.w16_plane = w16_plane; a.math16 = generic_bf16x3 ? 1 : 0; a.npl = 2;
    a.A = big_in; a.W = W; a.Out = small_out; a.xf = xf; a.ep = ep; a.d = d;
    a.M = d.N * d.HS * d.WS; a.CA = d.CB; a.Nn = d.CS;
    a.lws = ilog2_exact(d.WS); a.lhs = ilog2_exact(d.HS); a.dbgbuf = nullptr;
    a.sk_counter = nullptr; a.out_final = nullptr;
    GemmPlan p = plan_gemm(d, true, Wpacked != nullptr || Wp16 != nullptr, ws.ptr ? ws.floats : 0, (Wp16 && ws.counters) ? ws.ncounters : 0);
    if (planes16 == 3) {
        if (x6_route(d, true, p)) a.npl = 3;
        else {      // bf16x6 handle, launch outside the three-plane kernels: the exact-fp32 route
            if (!Wpacked && d.KS == 5) { fprintf(stderr, "uad: a bf16x6 launch outside the three-plane spatial kernels needs the fp32 pack (uad_conv_x6_takes / uad_conv_k3_takes)\n"); abort(); }
            a.Wp16 = nullptr;
            p = plan_gemm(d, true, Wpacked != nullptr, ws.ptr ? ws.floats : 0, 0);
        }
    }
    if (p.inkernel) a.sk_counter = ws.counters;
    run_plan(p, a, true, ws.ptr, st);
}

void uad_launch_conv_d(const UadConvDesc& d, const float* small_in, UadXform xf, const float* W, float* big_out,
                       UadEpilogue ep, hipStream_t st, const float* Wpacked, UadGemmWs ws, const unsigned short* Wp16,
                       long long w16_plane, bool generic_bf16x3, int planes16) {
    if (convk16_takes(d, false, xf, ep, Wp16)) { run_convk16(d, false, small_in, big_out, ep, Wp16, w16_plane, planes16, st); return; }
    ConvGemmArgs a;
    a.Wp = Wpacked; a.Wp16 = Wp16; a.w16_plane = w16_plane; a.math16 = generic_bf16x3 ? 1 : 0; a.npl = 2;
    a.A = small_in; a.W = W; a.Out = big_out; a.xf = xf; a.ep = ep; a.d = d;
    a.M = d.N * d.HS * d.WS; a.CA = d.CS; a.Nn = d.CB;
    a.lws = ilog2_exact(d.WS); a.lhs = ilog2_exact(d.HS); a.dbgbuf = nullptr;
    a.sk_counter = nullptr; a.out_final = nullptr;
    GemmPlan p = plan_gemm(d, false, Wpacked != nullptr || Wp16 != nullptr, ws.ptr ? ws.floats : 0, (Wp16 && ws.counters) ? ws.ncounters : 0);
    if (planes16 == 3) {
        // (the lane = pixel kernel's instance list: conv5_d16s_takes)
        const bool has_xf = xf.scale != nullptr;
        const bool inst_ok = ep.kind == UAD_EPI_FINAL ? (ep.fin_dc == nullptr && ep.ealpha >= 0.f && ep.ealpha <= 1.f && has_xf) : ep.kind == UAD_EPI_BWD_ACT ? !has_xf : has_xf;
        if (x6_route(d, false, p) && inst_ok) a.npl = 3;
        else {
            if (!Wpacked && d.KS == 5) { fprintf(stderr, "uad: a bf16x6 launch outside the three-plane spatial kernels needs the fp32 pack (uad_conv_x6_takes / uad_conv_k3_takes)\n"); abort(); }
            a.Wp16 = nullptr;
            p = plan_gemm(d, false, Wpacked != nullptr, ws.ptr ? ws.floats : 0, 0);
        }
    }
    if (p.inkernel) a.sk_counter = ws.counters;
    run_plan(p, a, false, ws.ptr, st);
}

// ---- W-type host side -------------------------------------------------------------------------
namespace {
struct WChoice { int BM, BN, splits, kper; };
inline WChoice choose_w(const UadConvDesc& d) {
    WChoice c;
    const int Mtot = d.KS * d.KS * d.CB;
    const long Kt = (long)d.N * d.HS * d.WS;
    c.BM = (Mtot >= 128) ? 128 : 64;
    c.BN = (d.CS > 32) ? 64 : 32;
    const long tiles = (long)((Mtot + c.BM - 1) / c.BM) * ((d.CS + c.BN - 1) / c.BN);
    const long ksteps = (Kt + 31) / 32;
    long splits = (1024 + tiles - 1) / tiles;          // aim for >= ~1024 workgroups
    long max_splits = (ksteps + 7) / 8;                // but keep >= 8 K-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long steps_per = (ksteps + splits - 1) / splits;
    c.kper = (int)steps_per * 32;
    c.splits = (int)((Kt + c.kper - 1) / c.kper);
    return c;
}
}  // namespace

namespace {
template <int NCSB, bool FBB, bool XFA, bool XFS, int NPL>
void launch_w_tr_n(const ConvWArgs& a, dim3 grid, const W5Choice& w5, hipStream_t st) {
    if constexpr (NCSB == 2 && NPL == 2) {
        // double-buffered form: 2 x 76 KB of LDS, one eight-wave workgroup per CU as before (UAD_NO_W5_DB: the single-buffered loop)
        static const bool nodb = getenv("UAD_NO_W5_DB") != nullptr;
        if (!nodb) {
            constexpr size_t ldsd = 2 * conv5_w_bf16_tr_lds_bytes(NCSB, NPL);
            static bool attrd = false;
            if (!attrd) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_w_bf16_tr_kernel<NCSB, FBB, XFA, XFS, NPL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd); attrd = true; }
            UAD_W_LAUNCH((conv5_w_bf16_tr_kernel<NCSB, FBB, XFA, XFS, NPL, true>), grid, dim3(256 * NCSB), ldsd, st, a, w5.tiles_per_split, w5.total_tiles);
            return;
        }
    }
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_w_bf16_tr_kernel<NCSB, FBB, XFA, XFS, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv5_w_bf16_tr_lds_bytes(NCSB, NPL)); attr = true; }
    UAD_W_LAUNCH((conv5_w_bf16_tr_kernel<NCSB, FBB, XFA, XFS, NPL>), grid, dim3(256 * NCSB), conv5_w_bf16_tr_lds_bytes(NCSB, NPL), st, a, w5.tiles_per_split, w5.total_tiles);
}
// bf16x6 instances: the three operand forms the model's backward uses -- (pattern word | plain gradient, activated input) in the decoder, (activated input,
// plain gradient) in the encoder; w_tr_x6_takes() tells the launcher, other forms of a bf16x6 launch run on the exact-fp32 generic kernels
inline bool w_tr_x6_takes(bool fbb, bool xa, bool xs) { return fbb ? xs : (xa != xs); }
template <int NCSB, bool FBB, bool XFA, bool XFS>
void launch_w_tr(const ConvWArgs& a, dim3 grid, const W5Choice& w5, hipStream_t st) {
    if constexpr (FBB ? XFS : (XFA != XFS)) { if (a.npl == 3) { launch_w_tr_n<NCSB, FBB, XFA, XFS, 3>(a, grid, w5, st); return; } }
    launch_w_tr_n<NCSB, FBB, XFA, XFS, 2>(a, grid, w5, st);
}
}  // namespace

bool uad_conv_w_supports_fb_bits(const UadConvDesc& d, bool math_bf16x3) {
    return math_bf16x3 && choose_w5(d).ok && d.CB == 32;
}

size_t uad_conv_w_partial_floats(const UadConvDesc& d) {
    const W5Choice w5 = choose_w5(d, false), w5b = choose_w5(d, true);      // (the launch's math mode picks one of the two: size for the larger)
    if (w5.ok) return (size_t)(w5.splits > w5b.splits ? w5.splits : w5b.splits) * d.KS * d.KS * d.CB * d.CS;
    const WChoice c = choose_w(d);
    const WK3Choice k3 = choose_wk3(d);      // (which of the two runs depends on the math mode of the launch: size for both)
    const int splits = (k3.ok && k3.splits > c.splits) ? k3.splits : c.splits;
    return (size_t)splits * d.KS * d.KS * d.CB * d.CS;
}

void uad_launch_conv_w(const UadConvDesc& d, const float* big, UadXform xfb, const float* small, UadXform xfs,
                       float* dW, float* partial, hipStream_t st, bool math_bf16x3, hipStream_t reduce_st, hipEvent_t ev, bool generic_bf16x3, bool defer_reduce, int planes16) {
    // planes16 == 3 (with math_bf16x3): bf16x6 products in the k5 s2 kernel; a shape that kernel does not take runs on the exact-fp32 generic kernels
    // reduce_st/ev (optional): run the split-K slab reduction on a second stream, ordered after the main kernel by `ev`.
    // defer_reduce: leave the reduction to a later uad_launch_conv_w_reduce (the caller orders it after this kernel with ONE event per layer:
    // every event recorded on the main stream costs a ~6 us bubble before its next kernel)
    auto hop = [&]() -> hipStream_t {
        if (!reduce_st) return st;
        (void)hipEventRecord(ev, st);
        (void)hipStreamWaitEvent(reduce_st, ev, 0);
        return reduce_st;
    };
    const W5Choice w5 = choose_w5(d, math_bf16x3);
    bool use16 = math_bf16x3;
    if (w5.ok && math_bf16x3 && planes16 == 3 && !w_tr_x6_takes(xfb.fb_bits != nullptr, xfb.scale != nullptr && !xfb.fb_bits, xfs.scale != nullptr)) {
        if (xfb.fb_bits) { fprintf(stderr, "uad: bf16x6 filter gradient from the pattern word needs the small operand's activation on load\n"); abort(); }
        use16 = false;      // the exact-fp32 k5 kernel on the SAME split (uad_launch_conv_w_reduce is told the launch's math mode, not the kernel)
    }
    if (w5.ok) {
        ConvWArgs a;
        a.big = big; a.small_ = small; a.partial = (w5.splits == 1) ? dW : partial;
        a.xfb = xfb; a.xfs = xfs; a.d = d;
        a.Mtot = d.KS * d.KS * d.CB; a.Kt = d.N * d.HS * d.WS; a.kper = 0; a.lws = a.lhs = -1; a.dbgbuf = nullptr;
        { static const int abl = getenv("UAD_W_ABL") ? atoi(getenv("UAD_W_ABL")) : 0; a.abl = abl; }
        a.npl = (use16 && planes16 == 3) ? 3 : 2;
        dim3 grid(d.CB / 32, d.CS / 32, w5.splits);
        static const int dbgw = getenv("UAD_DBG") ? atoi(getenv("UAD_DBG")) : 0;
        static unsigned long long* wbuf = nullptr;
        static int wcalls = 0;
        const bool dbg_this = (dbgw & 32) && wcalls < 8;
        constexpr size_t kWbuf = (size_t)2 * 2048 * 8;
        if (dbg_this) { if (!wbuf) (void)hipMalloc((void**)&wbuf, kWbuf); (void)hipMemsetAsync(wbuf, 0, kWbuf, st); a.dbgbuf = wbuf; }
        struct Dump { bool on; hipStream_t st; unsigned long long* buf; dim3* g; int* calls; UadConvDesc d;
            ~Dump() { if (!on) return; (void)hipStreamSynchronize(st); const int nb = g->x * g->y * g->z; static unsigned long long h[2 * 2048];
                (void)hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost); ++*calls;
                unsigned long long t0 = ~0ull, t1 = 0, dmin = ~0ull, dmax = 0, dsum = 0, smax = 0;
                for (int b = 0; b < nb && b < 2048; ++b) { if (h[2 * b] < t0) t0 = h[2 * b]; if (h[2 * b + 1] > t1) t1 = h[2 * b + 1]; }
                for (int b = 0; b < nb && b < 2048; ++b) { const unsigned long long dd = h[2 * b + 1] - h[2 * b]; dsum += dd; if (dd < dmin) dmin = dd; if (dd > dmax) dmax = dd; if (h[2 * b] - t0 > smax) smax = h[2 * b] - t0; }
                fprintf(stderr, "[w5 CB=%d CS=%d HS=%d grid=%d,%d,%d] span=%llu (100MHz ticks) wg dur min=%llu avg=%llu max=%llu latest start=%llu\n", d.CB, d.CS, d.HS, g->x, g->y, g->z,
                        t1 - t0, dmin, dsum / nb, dmax, smax); } } dump{dbg_this, st, wbuf, &grid, &wcalls, d};
        if (use16) {
            const bool two = d.CS % 64 == 0;       // two 32-channel cs blocks per workgroup (512 threads) where the layer has them
            if (two) grid.y = d.CS / 64;
            {
                const bool xa = xfb.scale != nullptr && !xfb.fb_bits, xs = xfs.scale != nullptr;
                // <cs blocks per workgroup, big operand from the pattern word, activation on load of big, ... of small>
                if (xfb.fb_bits) { if (two) { if (xs) launch_w_tr<2, true, false, true>(a, grid, w5, st); else launch_w_tr<2, true, false, false>(a, grid, w5, st); }
                                   else     { if (xs) launch_w_tr<1, true, false, true>(a, grid, w5, st); else launch_w_tr<1, true, false, false>(a, grid, w5, st); } }
                else if (two) { if (xa) { if (xs) launch_w_tr<2, false, true, true>(a, grid, w5, st); else launch_w_tr<2, false, true, false>(a, grid, w5, st); }
                                else    { if (xs) launch_w_tr<2, false, false, true>(a, grid, w5, st); else launch_w_tr<2, false, false, false>(a, grid, w5, st); } }
                else          { if (xa) { if (xs) launch_w_tr<1, false, true, true>(a, grid, w5, st); else launch_w_tr<1, false, true, false>(a, grid, w5, st); }
                                else    { if (xs) launch_w_tr<1, false, false, true>(a, grid, w5, st); else launch_w_tr<1, false, false, false>(a, grid, w5, st); } }
            }
        } else {
            hipLaunchKernelGGL(conv5_w_kernel, grid, dim3(256), 0, st, a, w5.tiles_per_split, w5.total_tiles);
        }
        if (w5.splits > 1 && !defer_reduce) uad_launch_reduce_partials(partial, w5.splits, a.Mtot * d.CS, 1.0f, dW, hop());
        return;
    }
    const WK3Choice k3 = choose_wk3(d);
    if (k3.ok && (math_bf16x3 || generic_bf16x3) && planes16 != 3 && !defer_reduce && !xfb.scale && !xfs.scale && !xfb.fb_bits) {
        // k3 s1 / s2 filter gradient in bf16x3 (uad_convk16.inc): channel-major LDS tiles, slabs + fixed-order reduction
        ConvWArgs a;
        a.big = big; a.small_ = small; a.partial = (k3.splits == 1) ? dW : partial;
        a.xfb = xfb; a.xfs = xfs; a.d = d;
        a.Mtot = 9 * d.CB; a.Kt = d.N * d.HS * d.WS; a.kper = 0; a.lws = a.lhs = -1; a.dbgbuf = nullptr;
        if (d.S == 1) { if (k3.ncb == 2) launch_convk_w16<1, 2>(a, k3, st); else launch_convk_w16<1, 1>(a, k3, st); }
        else launch_convk_w16<2, 1>(a, k3, st);
        if (k3.splits > 1) uad_launch_reduce_partials(partial, k3.splits, a.Mtot * d.CS, 1.0f, dW, hop());
        return;
    }
    const WChoice c = choose_w(d);
    ConvWArgs a;
    a.big = big; a.small_ = small; a.partial = (c.splits == 1) ? dW : partial;
    a.xfb = xfb; a.xfs = xfs; a.d = d;
    a.Mtot = d.KS * d.KS * d.CB; a.Kt = d.N * d.HS * d.WS; a.kper = c.kper;
    a.lws = ilog2_exact(d.WS); a.lhs = ilog2_exact(d.HS); a.dbgbuf = nullptr;
    dim3 grid((a.Mtot + c.BM - 1) / c.BM, (d.CS + c.BN - 1) / c.BN, c.splits);
    static const bool w16_ok = !getenv("UAD_NO_W16");
    const bool w16 = generic_bf16x3 && w16_ok && !xfb.scale && !xfs.scale && c.BN == 64 && d.CB % 4 == 0 && d.CS % 4 == 0;
    if (w16 && c.BM == 128)
        hipLaunchKernelGGL((conv_w16_kernel<128, 64, 2, 2>), grid, dim3(256), 0, st, a);
    else if (w16 && c.BM == 64)
        hipLaunchKernelGGL((conv_w16_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, a);
    else if (c.BM == 128 && c.BN == 64)
        hipLaunchKernelGGL((conv_w_kernel<128, 64, 32, 2, 2>), grid, dim3(256), 0, st, a);
    else if (c.BM == 128 && c.BN == 32)
        hipLaunchKernelGGL((conv_w_kernel<128, 32, 32, 4, 1>), grid, dim3(256), 0, st, a);
    else if (c.BM == 64 && c.BN == 64)
        hipLaunchKernelGGL((conv_w_kernel<64, 64, 32, 2, 2>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_w_kernel<64, 32, 32, 2, 1>), grid, dim3(128), 0, st, a);
    if (c.splits > 1 && !defer_reduce) uad_launch_reduce_partials(partial, c.splits, a.Mtot * d.CS, 1.0f, dW, hop());
}

void uad_conv_any_order_next(bool on) { g_any_order_next = on; }
void uad_conv_w_any_order_next(bool on) { g_any_order_w_next = on; }
void uad_launch_conv_w_reduce(const UadConvDesc& d, float* dW, float* partial, hipStream_t st, bool math_bf16x3) {
    const W5Choice w5 = choose_w5(d, math_bf16x3);
    const int splits = w5.ok ? w5.splits : choose_w(d).splits;
    if (splits > 1) uad_launch_reduce_partials(partial, splits, d.KS * d.KS * d.CB * d.CS, 1.0f, dW, st);
}
